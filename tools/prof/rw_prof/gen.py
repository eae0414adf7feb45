# instrumented copy (written to tools/scratch/, built by tools/prof/build.sh) of roma_amd/csrc/refiner_wide.hip: s_memtime stamps of every wave of workgroup 0 around the parts of phase kp = 8
import os
here = os.path.dirname(os.path.abspath(__file__))
out_dir = os.path.join(here, "../../scratch", os.path.basename(here))   # generated source + library: tools/scratch/ (git-ignored, travels to the box)
os.makedirs(out_dir, exist_ok=True)
s = open(os.path.join(here, "../../../roma_amd/csrc/refiner_wide.hip")).read()
def rep(a, b):
    global s
    assert s.count(a) == 1, (s.count(a), a[:70])
    s = s.replace(a, b)
rep('#include "common.h"\n#include "lc_device.h"', '#include "../../../roma_amd/csrc/common.h"\n#include "../../../roma_amd/csrc/lc_device.h"\n__device__ unsigned long long g_prof[8 * 16];\n#define PROF(i) do { if (blockIdx.x == 300 && kp == 8 && (threadIdx.x & 63) == 0) g_prof[(threadIdx.x >> 6) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)')
rep("""    constexpr int buf = decltype(buf_c)::value;                 // t / weight buffer of panel kp; panel kp + 1 uses the other ones
    if (kp + 2 < NKP) load_x(kp + 2);""", """    constexpr int buf = decltype(buf_c)::value;                 // t / weight buffer of panel kp; panel kp + 1 uses the other ones
    PROF(0);
    if (kp + 2 < NKP) load_x(kp + 2);
    PROF(1);""")
rep("""    if (dw_first && kp + 1 < NKP) dw_panel(kp + 1, buf ^ 1);
    mfma_panel(buf, !dw_first && kp + 1 < NKP ? kp + 1 : -1);
    if (!dw_first && kp + 1 < NKP) dw_panel(kp + 1, buf ^ 1);
    if (kp + 2 < NKP) store_x(buf);                             // halo + taps of panel kp + 2 -> X buffer buf (panel kp's: read a phase ago)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the next panel's weights have landed
    __syncthreads();""", """    if (dw_first && kp + 1 < NKP) dw_panel(kp + 1, buf ^ 1);
    PROF(2);
    mfma_panel(buf, !dw_first && kp + 1 < NKP ? kp + 1 : -1);
    PROF(3);
    if (!dw_first && kp + 1 < NKP) dw_panel(kp + 1, buf ^ 1);
    PROF(4);
    if (kp + 2 < NKP) store_x(buf);                             // halo + taps of panel kp + 2 -> X buffer buf (panel kp's: read a phase ago)
    PROF(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the next panel's weights have landed
    PROF(6);
    __syncthreads();
    PROF(7);""")
s += '\nextern "C" int rw_prof_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 128); }\n'
open(os.path.join(out_dir, "rw_prof.hip"), "w").write(s)
