// Sanitizer build of the library's host-side arithmetic: g++ -fsanitize=address,undefined.
// Exercises tnml_trunc_rank over a grid of shapes / policies and the canonical <-> sweep-relative layout round trip;
// exits 0 when every invariant holds (and the sanitizers stayed silent).
#include <cstdio>
#include <vector>

#include "host_plan.inc"

int main() {
  long checked = 0;
  for (int policy = 0; policy < 3; ++policy)
    for (int left = 0; left < 2; ++left)
      for (int N = 2; N <= 9; ++N)
        for (int p = 0; p <= N - 2; ++p)
          for (int ml = 1; ml <= 7; ++ml)
            for (int mr = 1; mr <= 7; ++mr)
              for (int L = 1; L <= 10; L += 3)
                for (int M = 1; M <= 9; M += 2) {
                  const int D = 2;
                  const int m = tnml_trunc_rank(policy, left, p, N, ml, D, mr, L, M);
                  const int rows = left ? D * ml * L : D * ml, cols = left ? D * mr : D * mr * L;
                  const int nS = std::min(rows, cols);
                  if (m == TNML_ERR_SHAPE) { if (policy != TNML_TRUNC_REFERENCE) return 1; continue; }
                  if (m < 1 || m > nS) return 2;
                  if (policy != TNML_TRUNC_REFERENCE && m != std::min(M, nS)) return 3;
                  ++checked;
                }
  for (int left = 0; left < 2; ++left) {
    const int ml = 3, mr = 5, D = 2, L = 4;
    const size_t n = (size_t)ml * D * D * mr * L;
    std::vector<float> src(n), rel(n);
    std::vector<double> seen(n, 0.0);
    for (size_t e = 0; e < n; ++e) src[e] = (float)e;
    canon_to_rel(src.data(), rel.data(), left, ml, mr, D, L);
    for (size_t e = 0; e < n; ++e) seen[(size_t)rel[e]] += 1.0;      // a permutation: every source element exactly once
    for (size_t e = 0; e < n; ++e) if (seen[e] != 1.0) return 4;
  }
  std::printf("host_plan sanitizer test ok (%ld shapes)\n", checked);
  return 0;
}
