/* tests/oracle_sanitize_driver.c -- the CPU oracle (oracle/spamat_oracle.c, TEST INFRASTRUCTURE) under AddressSanitizer +
 * UndefinedBehaviorSanitizer: every entry on edge shapes (W < max_disp, W = 1, max_disp = 1, empty / full / ragged masks,
 * one channel) with buffers of EXACTLY the documented sizes on the heap, so an index outside a plane is a sanitizer report.
 * Built and run by tests/test_oracle_sanitizers.py:  gcc -fsanitize=address,undefined ... driver.c spamat_oracle.c */
#include <stdio.h>
#include <stdlib.h>

int oracle_spamat_forward(const float *, const float *, const float *, const float *, float *, float *, float *, int, int,
                          int, int, int);
int oracle_spamat_backward(const float *, const float *, const float *, const float *, const float *, const float *,
                           const float *, const float *, float *, float *, int, int, int, int, int);
int oracle_spavar_forward(const float *, const float *, const float *, const float *, const float *, float *, float *,
                          float *, int, int, int, int, int);
int oracle_spavar_backward(const float *, const float *, const float *, const float *, const float *, const float *,
                           const float *, const float *, const float *, float *, float *, float *, int, int, int, int, int);

static unsigned long long s = 88172645463325252ull;
static float rnd(void) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (float)((s >> 40) & 0xffff) / 65536.0f;
}
static float *buf(size_t n, int mode, float p) {          /* mode 0: features, 1: mask of density p, 2: output */
    float *b = (float *)malloc(n * sizeof(float));
    if (!b) exit(2);
    for (size_t i = 0; i < n; ++i) b[i] = mode == 0 ? rnd() * 2.0f - 0.5f : mode == 1 ? (rnd() < p ? 1.0f : 0.0f) : -7.0f;
    return b;
}

static int one(int B, int C, int H, int W, int D, float p) {
    const size_t nf = (size_t)B * C * H * W, np = (size_t)B * H * W;
    float *L = buf(nf, 0, 0), *R = buf(nf, 0, 0), *rm = buf(np, 1, p), *tm = buf(np, 1, p), *g = buf(np, 0, 0);
    float *o = buf(np, 2, 0), *ss = buf(np, 2, 0), *mx = buf(np, 2, 0), *v = buf(np, 2, 0), *s2 = buf(np, 2, 0),
          *m2 = buf(np, 2, 0), *gl = buf(nf, 2, 0), *gr = buf(nf, 2, 0), *gd = buf(np, 2, 0);
    int ok = oracle_spamat_forward(L, R, rm, tm, o, ss, mx, B, C, H, W, D) == 1;
    ok &= oracle_spamat_backward(L, R, rm, tm, o, ss, mx, g, gl, gr, B, C, H, W, D) == 1;
    ok &= oracle_spavar_forward(L, R, rm, tm, o, v, s2, m2, B, C, H, W, D) == 1;
    ok &= oracle_spavar_backward(L, R, rm, tm, o, v, s2, m2, g, gl, gr, gd, B, C, H, W, D) == 1;
    for (size_t i = 0; i < np; ++i) {                    /* masked-off pixels: the forward leaves what the caller wrote */
        const float hi = D - 1 > 1 ? (float)(D - 1) : 1.0f;   /* an empty candidate set gives 1e-6 / 1e-6 = 1 (SM_kernel.cu:121) */
        if (rm[i] != 0.0f && !(o[i] >= 0.0f && o[i] <= hi + 1e-3f)) ok = 0;
    }
    free(L); free(R); free(rm); free(tm); free(g); free(o); free(ss); free(mx); free(v); free(s2); free(m2);
    free(gl); free(gr); free(gd);
    return ok;
}

int main(void) {
    static const int shapes[][5] = {{1, 1, 1, 1, 1},  {1, 1, 1, 1, 9},   {2, 3, 2, 5, 8},  {1, 8, 3, 40, 24}, {1, 2, 1, 7, 1},
                                    {1, 4, 2, 33, 33}, {1, 4, 2, 33, 64}, {3, 1, 4, 17, 5}, {1, 24, 2, 20, 72}};
    static const float dens[] = {0.0f, 0.3f, 1.0f};
    int n = 0;
    for (unsigned i = 0; i < sizeof(shapes) / sizeof(shapes[0]); ++i)
        for (unsigned k = 0; k < 3; ++k, ++n)
            if (!one(shapes[i][0], shapes[i][1], shapes[i][2], shapes[i][3], shapes[i][4], dens[k])) {
                printf("FAILED shape %u density %g\n", i, dens[k]);
                return 1;
            }
    printf("SANITIZED_OK %d cases\n", n);
    return 0;
}
