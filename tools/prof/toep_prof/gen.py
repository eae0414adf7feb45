# instrumented copy (written to tools/scratch/, built by tools/prof/build.sh) of roma_amd/csrc/refiner_block.hip: s_memtime stamps of the four waves of workgroup 100 in its third tile
import os
here = os.path.dirname(os.path.abspath(__file__))
out_dir = os.path.join(here, "../../scratch", os.path.basename(here))   # generated source + library: tools/scratch/ (git-ignored, travels to the box)
os.makedirs(out_dir, exist_ok=True)
s = open(os.path.join(here, "../../../roma_amd/csrc/refiner_block.hip")).read()
def rep(a, b, cnt=1):
    global s
    assert s.count(a) == cnt, (s.count(a), a[:70])
    s = s.replace(a, b)
rep('#include "common.h"\n#include "lc_device.h"', '#include "../../../roma_amd/csrc/common.h"\n#include "../../../roma_amd/csrc/lc_device.h"\n__device__ unsigned long long g_prof[4 * 16];\n#define PROF(i) do { if (blockIdx.x == 100 && tile == 100 + 2 * (int)gridDim.x && (threadIdx.x & 63) == 0) g_prof[(threadIdx.x >> 6) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)')
rep("    const bool more = tile + (int)gridDim.x < ntile, more2 = tile + 2 * (int)gridDim.x < ntile;\n", "    const bool more = tile + (int)gridDim.x < ntile, more2 = tile + 2 * (int)gridDim.x < ntile;\n    PROF(0);\n    PROF(1);\n")
rep("    __syncthreads();\n    // ---- 1x1 on the matrix cores", "    PROF(2);\n    __syncthreads();\n    PROF(3);\n    // ---- 1x1 on the matrix cores")
rep("    __syncthreads();                                             // every wave is done reading t", "    PROF(4);\n    __syncthreads();\n    PROF(5);")
rep("    if (more) scatter();\n    if (more2) fetch(tile + 2 * gridDim.x);\n    __syncthreads();", "    PROF(6);\n    if (more) scatter();\n    if (more2) fetch(tile + 2 * gridDim.x);\n    PROF(7);\n    __syncthreads();\n    PROF(8);")
rep("    __syncthreads();                                             // staging read; t's fourth packet must be zero again before the next depthwise", "    PROF(9);\n    __syncthreads();\n    PROF(10);")
rep("    // (no barrier needed here: the depthwise writes other slots of t", "    PROF(11);\n    // (no barrier needed here: the depthwise writes other slots of t")
s += '\nextern "C" int toep_prof_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 64); }\n'
open(os.path.join(out_dir, "toep_prof.hip"), "w").write(s)
