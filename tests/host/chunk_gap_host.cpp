// Host driver of the chunk boundary check (csrc/dfm_chunk_core.h: state_gap / gap_close): the text recursion_chunk_kernel's lanes
// run when they compare the state a chunk starts from with the state the neighbouring chunk ends on.  TEST INFRASTRUCTURE ONLY
// (tests/test_chunk_core_cpu.py).
// stdin (binary): int n; then n records of { double tol, m[36], x[8], m2[36], x2[8] };  stdout (text): one 0 / 1 per record
#include <cstdio>
#include "../../dynamic_factor_models_amd/csrc/dfm_chunk_core.h"
using namespace dfm::chunk;

int main() {
    int n;
    if (fread(&n, sizeof(int), 1, stdin) != 1) return 1;
    for (int k = 0; k < n; ++k) {
        double tol, m[NP], x[R], m2[NP], x2[R];
        if (fread(&tol, sizeof(double), 1, stdin) != 1 || fread(m, sizeof(double), NP, stdin) != NP || fread(x, sizeof(double), R, stdin) != R ||
            fread(m2, sizeof(double), NP, stdin) != NP || fread(x2, sizeof(double), R, stdin) != R) return 2;
        printf("%d\n", gap_close(state_gap(m, x, m2, x2), tol) ? 1 : 0);
    }
    return 0;
}
