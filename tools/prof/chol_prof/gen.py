# instrumented copy (written to tools/scratch/, built by tools/prof/build.sh) of roma_amd/csrc/chol.hip (100 MHz wall-clock stamps in the diagonal workgroup of chol_step_kernel)
import os
here = os.path.dirname(os.path.abspath(__file__))
out_dir = os.path.join(here, "../../scratch", os.path.basename(here))   # generated source + library: tools/scratch/ (git-ignored, travels to the box)
os.makedirs(out_dir, exist_ok=True)
s = open(os.path.join(here, "../../../roma_amd/csrc/chol.hip")).read()
def rep(a, b):
    global s
    assert s.count(a) == 1, (s.count(a), a[:70])
    s = s.replace(a, b)
rep('#include "common.h"', '#include "../../../roma_amd/csrc/common.h"\n__device__ unsigned long long g_prof[16];\n#define PROF(i) do { if (prof && threadIdx.x == 0) g_prof[i] = wall_clock64(); } while (0)')
rep("""  const int tid = threadIdx.x, b = blockIdx.y;
  const int e = p.j + p.nb;
  // tile id""", """  const int tid = threadIdx.x, b = blockIdx.y;
  const int e = p.j + p.nb;
  const bool prof = blockIdx.x == p.ntc && b == 0;
  PROF(0);
  // tile id""")
rep("""  __syncthreads();
  auto panel""", """  __syncthreads();
  PROF(1);
  auto panel""")
rep("""  if (I >= 0 && I != J) panel(sI, ri);
  __syncthreads();""", """  if (I >= 0 && I != J) panel(sI, ri);
  __syncthreads();
  PROF(2);""")
rep("  if (!next_diag) return;", "  PROF(3);\n  if (!next_diag) return;")
rep("""  chol_factor_lds(Ls, Ws, Ts, s_inv, s_bad, tid);
  float* Wnb""", """  PROF(4);
  chol_factor_lds(Ls, Ws, Ts, s_inv, s_bad, tid, prof);
  float* Wnb""")
rep("float* __restrict__ s_inv, int* __restrict__ s_bad_p, int tid) {", "float* __restrict__ s_inv, int* __restrict__ s_bad_p, int tid, bool prof = false) {")
rep("""    __syncthreads();
    // (2) rows below the diagonal block""", """    __syncthreads();
    if (p == 0) PROF(11);
    if (p == 1) PROF(14);
    // (2) rows below the diagonal block""")
rep("""    __syncthreads();
    // (3) trailing lower triangle""", """    __syncthreads();
    if (p == 0) PROF(12);
    if (p == 1) PROF(15);
    // (3) trailing lower triangle""")
rep("""      __syncthreads();
    }
  }

  // (4a) diagonal blocks of W""", """      __syncthreads();
    }
    if (p == 0) PROF(13);
  }

  PROF(5);
  // (4a) diagonal blocks of W""")
rep("""  __syncthreads();
  // off-diagonal block (rows r0""", """  __syncthreads();
  PROF(6);
  // off-diagonal block (rows r0""")
rep("""  offdiag(2 * PB, 0, std::integral_constant<int, 2 * PB>{});    // (4c) the 32 x 32 block below the diagonal
""", """  PROF(7);
  offdiag(2 * PB, 0, std::integral_constant<int, 2 * PB>{});    // (4c) the 32 x 32 block below the diagonal
  PROF(8);
""")
rep("""  if (tid == 0 && *s_bad != 0 && *s_bad <= nbn && p.info[b] == 0) p.info[b] = p.info_base + *s_bad;
}""", """  if (tid == 0 && *s_bad != 0 && *s_bad <= nbn && p.info[b] == 0) p.info[b] = p.info_base + *s_bad;
  PROF(10);
}""")
s += '\nextern "C" int chol_prof_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 16); }\n'
open(os.path.join(out_dir, "chol_prof.hip"), "w").write(s)
