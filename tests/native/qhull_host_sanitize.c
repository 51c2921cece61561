/* AddressSanitizer + UBSan driver for csrc/mvosr_qhull_host.c (CPU build only: GPU sanitizers are not available on this pool): 3 000
 * point sets — random, quarter-pixel grid, coarse grid with duplicates and collinear rows, collinear, duplicates, near-cocircular, NaN,
 * fewer than three points — plus a too-small rows buffer on every set.  Built and run by tests/test_qhull_host.py::test_sanitizers. */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
int mvosr_qhull_rows_host(const double *points, int64_t n_points, int64_t stride_doubles, int32_t *rows, int64_t rows_cap, int32_t *n_rows, int32_t *order);
static double u01(void) { return drand48(); }
int main(void) {
    srand48(12345);
    long ok = 0, declined = 0, err = 0;
    for (int rep = 0; rep < 3000; ++rep) {
        int kind = rep % 8;
        int n = kind == 7 ? (int)(u01() * 6) : 3 + (int)(u01() * (rep % 50 == 0 ? 5000 : 900));
        double *p = malloc(sizeof(double) * 2 * (n > 0 ? n : 1));
        int32_t *rows = malloc(sizeof(int32_t) * 3 * (2 * n + 8));
        int32_t *order = malloc(sizeof(int32_t) * (n > 0 ? n : 1));
        for (int i = 0; i < n; ++i) {
            double x = u01() * 1241.0, y = 186.0 + u01() * 190.0;
            if (kind == 1) { x = floor(x * 4) / 4; y = floor(y * 4) / 4; }            /* quarter-pixel grid */
            if (kind == 2) { x = floor(x); y = floor(y / 8) * 8; }                    /* coarse grid: duplicates, collinear rows */
            if (kind == 3) { y = 200.0 + 0.1 * x; }                                    /* collinear */
            if (kind == 4 && i % 3 == 0 && i) { x = p[2 * (i - 1)]; y = p[2 * (i - 1) + 1]; }   /* duplicates */
            if (kind == 5) { double a = 6.283185307179586 * i / n; x = 600 + 100 * cos(a); y = 280 + 90 * sin(a); }   /* cocircular-ish */
            if (kind == 6 && i == n / 2) { x = NAN; }
            p[2 * i] = x; p[2 * i + 1] = y;
        }
        int32_t nr = -1;
        int rc = mvosr_qhull_rows_host(p, n, 2, rows, 2 * n + 8, &nr, (rep & 1) ? order : NULL);
        if (rc == 0) { ok++; for (int t = 0; t < 3 * nr; ++t) if (rows[t] < 0 || rows[t] >= n) { printf("BAD ROW %d\n", rows[t]); return 1; } }
        else if (rc > 0) declined++; else err++;
        /* a too-small rows buffer must decline, not overflow */
        if (n >= 10) { rc = mvosr_qhull_rows_host(p, n, 2, rows, 5, &nr, NULL); if (rc == 0) { printf("rows_cap ignored\n"); return 1; } }
        free(p); free(rows); free(order);
    }
    printf("sanitizer run: ok %ld declined %ld errors %ld\n", ok, declined, err);
    return 0;
}
