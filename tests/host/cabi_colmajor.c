/* tests/host/cabi_colmajor.c -- the closest thing to EXECUTING julia/dfm_hip.jl this image allows (no Julia here): a C host that
 * holds its matrices exactly as the Julia shim's caller does -- COLUMN-major `T x ns` data with NaN where the reference has
 * `missing` (nan_for_missing, julia/dfm_hip.jl:57), column-major ns x r loadings, r x r transition matrices -- performs the shim's
 * own marshalling step by step (`to_c_panel` = permutedims(z, (2, 1)) at :56; `reshape(permutedims(p.Lam), r, N, 1)` at :82-84;
 * the `permutedims` back at :74-75 and :111-113; `flags = any(isnan, z) ? DFM_F_MAY_HAVE_MISSING : 0` at :88) and calls
 * dfm_pca_init_batch / dfm_em_batch through include/dfm_hip.h as the shim's ccall does (:67-71, :91-96).
 * tests/test_gpu_cabi_colmajor.py compiles it, feeds it column-major inputs and compares the column-major outputs with the oracle.
 * TEST INFRASTRUCTURE ONLY.
 *   usage: cabi_colmajor <in.bin> <out.bin>
 *   in : int mode (0 = pca_init, 1 = em), T, N, r, max_iter; double z[T x N col-major];
 *        mode 1: Lam[N x r col-major], R[N], A[r x r cm], Q[r x r cm], mu0[r], P0[r x r cm]
 *   out: mode 0: Lam[N x r cm], R[N], A, Q, mu0, P0 (cm), F[T x r cm]
 *        mode 1: int iters; Lam, R, A, Q, mu0, P0 (cm), loglik[max_iter], factor[T x r cm]                                      */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "dfm_hip.h"

/* Julia: permutedims(x, (2, 1)) of a column-major (rows x cols) matrix -> column-major (cols x rows) */
static double* permutedims21(const double* x, int rows, int cols) {
    double* y = (double*)malloc(sizeof(double) * (size_t)rows * cols);
    for (int c = 0; c < cols; ++c)
        for (int r = 0; r < rows; ++r) y[(size_t)r * cols + c] = x[(size_t)c * rows + r];   /* y[c, r] = x[r, c] */
    return y;
}
static double* rd(FILE* f, size_t n) {
    double* p = (double*)malloc(sizeof(double) * n);
    if (fread(p, sizeof(double), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}
static void wr(FILE* f, const double* p, size_t n) { fwrite(p, sizeof(double), n, f); }
static void fail(dfm_handle* h, int rc, const char* what) {
    fprintf(stderr, "%s: status %d: %s\n", what, rc, h ? dfm_last_error(h) : "no handle");
    exit(3);
}

int main(int argc, char** argv) {
    if (argc != 3) return 1;
    FILE* fi = fopen(argv[1], "rb");
    FILE* fo = fopen(argv[2], "wb");
    if (!fi || !fo) return 1;
    int hd[5];
    if (fread(hd, sizeof(int), 5, fi) != 5) return 2;
    const int mode = hd[0], T = hd[1], N = hd[2], r = hd[3], max_iter = hd[4];
    double* z = rd(fi, (size_t)T * N);                       /* T x N, column-major, NaN = missing */
    dfm_handle* h = NULL;
    int rc = dfm_create(&h, 0, NULL);
    if (rc) fail(h, rc, "dfm_create");
    double* panel = permutedims21(z, T, N);                  /* to_c_panel: (N, T, 1) column-major == C [b][t][i] */
    const int np = r * (r + 1) / 2;
    if (mode == 0) {
        double *Lam = malloc(sizeof(double) * r * N), *R = malloc(sizeof(double) * N), *A = malloc(sizeof(double) * r * r),
               *Q = malloc(sizeof(double) * r * r), *mu0 = malloc(sizeof(double) * r), *P0 = malloc(sizeof(double) * r * r),
               *F = malloc(sizeof(double) * (size_t)r * T);
        rc = dfm_pca_init_batch(h, 1, T, N, r, panel, Lam, R, A, Q, mu0, P0, F);
        if (rc) fail(h, rc, "dfm_pca_init_batch");
        /* Julia arrays (r, N, 1), (r, r, 1), (r, T, 1): permutedims(X[:, :, 1]) -> N x r, r x r, T x r (column-major) */
        double* LamJ = permutedims21(Lam, r, N); double* AJ = permutedims21(A, r, r); double* QJ = permutedims21(Q, r, r);
        double* P0J = permutedims21(P0, r, r); double* FJ = permutedims21(F, r, T);
        wr(fo, LamJ, (size_t)N * r); wr(fo, R, N); wr(fo, AJ, r * r); wr(fo, QJ, r * r); wr(fo, mu0, r); wr(fo, P0J, r * r); wr(fo, FJ, (size_t)T * r);
    } else {
        double* LamJ = rd(fi, (size_t)N * r); double* R = rd(fi, N); double* AJ = rd(fi, r * r); double* QJ = rd(fi, r * r);
        double* mu0 = rd(fi, r); double* P0J = rd(fi, r * r);
        double* Lam = permutedims21(LamJ, N, r);             /* reshape(permutedims(p.Lam), r, N, 1) */
        double* A = permutedims21(AJ, r, r); double* Q = permutedims21(QJ, r, r); double* P0 = permutedims21(P0J, r, r);
        double* path = malloc(sizeof(double) * max_iter); int iters = 0;
        double* f = malloc(sizeof(double) * (size_t)r * T); double* P = malloc(sizeof(double) * (size_t)np * T);
        unsigned flags = 0;
        for (size_t k = 0; k < (size_t)T * N; ++k)
            if (isnan(z[k])) { flags = DFM_F_MAY_HAVE_MISSING; break; }   /* any(isnan, z) */
        rc = dfm_em_batch(h, 1, T, N, r, panel, Lam, R, A, Q, mu0, P0, max_iter, 0.0, path, &iters, f, P, flags);
        if (rc) fail(h, rc, "dfm_em_batch");
        double* LamO = permutedims21(Lam, r, N); double* AO = permutedims21(A, r, r); double* QO = permutedims21(Q, r, r);
        double* P0O = permutedims21(P0, r, r); double* fO = permutedims21(f, r, T);
        fwrite(&iters, sizeof(int), 1, fo);
        wr(fo, LamO, (size_t)N * r); wr(fo, R, N); wr(fo, AO, r * r); wr(fo, QO, r * r); wr(fo, mu0, r); wr(fo, P0O, r * r);
        wr(fo, path, max_iter); wr(fo, fO, (size_t)T * r);
    }
    dfm_destroy(h);
    fclose(fi); fclose(fo);
    return 0;
}
