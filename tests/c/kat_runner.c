/* tests/c/kat_runner.c — the reference's known-answer cases through the reference's own C API
 * (include/audiosync/cross_correlation.h), linked against libaudiosync.so.  The cases are the
 * data of tests/golden/reference_kat.txt (from tests/test_cross_correlation.c:21-113 and
 * tests/test_pearson_coefficient.c:20-58 of the reference).  Exit code 0 = all good. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <audiosync/audiosync.h>
#include <audiosync/cross_correlation.h>

static int check(const char *name, char mode, double bound, double v)
{
    int ok = 1;
    if (mode == 'E') ok = (v == bound);
    else if (mode == 'G') ok = (v > bound);
    else if (mode == 'L') ok = (v < bound);
    if (!ok) fprintf(stderr, "FAIL %s: coefficient %.17g violates %c %.17g\n", name, v, mode, bound);
    return ok;
}

static double *read_values(FILE *f, size_t n)
{
    double *v = malloc(sizeof(double) * (n ? n : 1));
    for (size_t i = 0; i < n; i++)
        if (fscanf(f, "%lf", &v[i]) != 1) { free(v); return NULL; }
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s reference_kat.txt\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "r");
    if (!f) { perror("open"); return 2; }
    char kind, name[64], mode;
    int failures = 0, cases = 0;
    while (fscanf(f, " %c %63s", &kind, name) == 2) {
        if (kind == 'X') {
            size_t n; int want_ret; long want_lag; double bound;
            if (fscanf(f, "%zu %d %ld %c %lf", &n, &want_ret, &want_lag, &mode, &bound) != 5) return 2;
            double *source = read_values(f, 2 * n), *sample = read_values(f, n);
            if (!source || !sample) return 2;
            long lag = -777; double coef = -7.0;
            const int ret = cross_correlation(source, sample, n, &lag, &coef);
            printf(">> %s: returned %d lag=%ld coef=%f\n", name, ret, lag, coef);
            if (ret != want_ret) { fprintf(stderr, "FAIL %s: ret %d != %d\n", name, ret, want_ret); failures++; }
            else if (ret == 0) {
                if (lag != want_lag) { fprintf(stderr, "FAIL %s: lag %ld != %ld\n", name, lag, want_lag); failures++; }
                if (!check(name, mode, bound, coef)) failures++;
            }
            free(source); free(sample);
        } else if (kind == 'P') {
            size_t n; double bound;
            if (fscanf(f, "%zu %c %lf", &n, &mode, &bound) != 3) return 2;
            double *a = read_values(f, n), *b = read_values(f, n);
            if (!a || !b) return 2;
            const double v = pearson_coefficient(a, a + n, b, b + n);
            printf(">> %s: returned %f\n", name, v);
            if (mode == 'N') { if (v == v) { fprintf(stderr, "FAIL %s: expected NaN\n", name); failures++; } }
            else if (!check(name, mode, bound, v)) failures++;
            free(a); free(b);
        } else {
            return 2;
        }
        cases++;
    }
    fclose(f);
    printf("%d cases, %d failures\n", cases, failures);
    return failures ? 1 : (cases == 12 ? 0 : 2);
}
