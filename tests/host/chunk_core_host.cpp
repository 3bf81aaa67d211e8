// Host driver of csrc/dfm_chunk_core.h (the per-lane algebra of recursion_chunk.hip): ONE lane runs the whole sample
// sequentially -- forward steps, terminal state, backward steps with the EM accumulators -- so that the text the GPU lanes
// execute is checked against oracle/kalman_oracle.py on the CPU (tests/test_chunk_core_cpu.py).  TEST INFRASTRUCTURE ONLY.
// stdin (binary): int T; double K[64], Phi[64], QPhi[64], M0[36], xi0[8], C[T][36], b[T][8]
// stdout (binary): double ldsum, xwsum, ldT, xfT; P[T+1][36], f[T+1][8] (states 0..T); S10[64], S11[36]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../dynamic_factor_models_amd/csrc/dfm_chunk_core.h"
using namespace dfm::chunk;

struct Rows { const double* p; Row8 operator()(int i, const Deps&) const { Row8 r; for (int k = 0; k < R; ++k) r.v[k] = p[8 * i + k]; return r; } };
struct ObsH { const double* c_; const double* b_; void ready() const {} double c(int p) const { return c_[p]; } double b(int i) const { return b_[i]; } };
struct ZwH { const double* z_; const double* w_; void ready() const {} double z(int p) const { return z_[p]; } double w(int i) const { return w_[i]; } };
struct Acc {
    static constexpr bool on = true;
    Acc(double* a, double* b) : s10_(a), s11_(b) {}
    double *s10_, *s11_;
    bool want10() const { return true; }
    bool want11() const { return true; }
    void s10(int k, int n, double v) const { s10_[8 * k + n] += v; }
    void s11(int p, double v) const { s11_[p] += v; }
};

int main() {
    int T;
    if (fread(&T, sizeof(int), 1, stdin) != 1) return 1;
    std::vector<double> in(64 * 3 + 36 + 8 + (size_t)T * 44);
    if (fread(in.data(), sizeof(double), in.size(), stdin) != in.size()) return 2;
    const double *K = in.data(), *Phi = K + 64, *QPhi = Phi + 64, *M0 = QPhi + 64, *xi0 = M0 + 36, *C = xi0 + 8, *b = C + (size_t)T * 36;
    double KT[64];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) KT[8 * i + j] = K[8 * j + i];
    double m[NP], xi[R];
    for (int k = 0; k < NP; ++k) m[k] = M0[k];
    for (int i = 0; i < R; ++i) xi[i] = xi0[i];
    std::vector<double> Zn((size_t)T * 36), W((size_t)T * 8);
    double ldsum = 0.0, xwsum = 0.0;
    auto rcp = [](double d) { return 1.0 / d; };
    for (int t = 0; t < T; ++t) {
        double det, xw;
        fwd_step(m, xi, ObsH{C + (size_t)t * 36, b + (size_t)t * 8}, det, xw, Rows{K}, Rows{QPhi}, rcp, [&](const double (&zn)[NP], const double (&w)[R]) {
            for (int k = 0; k < NP; ++k) Zn[(size_t)t * 36 + k] = zn[k];
            for (int i = 0; i < R; ++i) W[(size_t)t * 8 + i] = w[i];
        });
        ldsum += log(det); xwsum += xw;
    }
    // terminal: Om_f,T = m - Phi; P_T = Om^-1; f_T = P_T xi
    double om[NP];
    for (int i = 0; i < R; ++i) for (int j = 0; j <= i; ++j) om[pidx(i, j)] = m[pidx(i, j)] - Phi[8 * i + j];
    const double detT = sweep8(om, rcp);
    double P[NP], f[R];
    for (int k = 0; k < NP; ++k) P[k] = -om[k];
    double xfT = 0.0;
    for (int i = 0; i < R; ++i) { double s = 0.0; for (int q = 0; q < R; ++q) s += P[pidx(i, q)] * xi[q]; f[i] = s; xfT += xi[i] * s; }
    std::vector<double> Ps((size_t)(T + 1) * 36), fs((size_t)(T + 1) * 8);
    double S10[64] = {0}, S11[36] = {0};
    for (int k = 0; k < NP; ++k) { Ps[(size_t)T * 36 + k] = P[k]; }
    for (int i = 0; i < R; ++i) fs[(size_t)T * 8 + i] = f[i];
    for (int i = 0; i < R; ++i) for (int j = 0; j <= i; ++j) S11[pidx(i, j)] += P[pidx(i, j)] + f[i] * f[j];
    for (int t = T - 1; t >= 0; --t) {
        double s11tmp[36] = {0};
#ifdef CHUNK_HOST_NOACC   // the plain pass's instantiation (f_t formed after G): S10 / S11 stay 0
        (void)s11tmp;
        bwd_step(P, f, ZwH{Zn.data() + (size_t)t * 36, W.data() + (size_t)t * 8}, Rows{KT}, NoAcc{});
#else
        bwd_step(P, f, ZwH{Zn.data() + (size_t)t * 36, W.data() + (size_t)t * 8}, Rows{KT}, Acc(S10, t > 0 ? S11 : s11tmp));
#endif
        for (int k = 0; k < NP; ++k) Ps[(size_t)t * 36 + k] = P[k];
        for (int i = 0; i < R; ++i) fs[(size_t)t * 8 + i] = f[i];
    }
    const double head[4] = {ldsum, xwsum, log(detT), xfT};
    fwrite(head, sizeof(double), 4, stdout);
    fwrite(Ps.data(), sizeof(double), Ps.size(), stdout);
    fwrite(fs.data(), sizeof(double), fs.size(), stdout);
    fwrite(S10, sizeof(double), 64, stdout);
    fwrite(S11, sizeof(double), 36, stdout);
    return 0;
}
