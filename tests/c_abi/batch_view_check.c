/* pa_batch_align_view from plain C (include/pa_bitpacking_hip.h): the batched A*PA2 of both presets on 48 pairs of 300-2400 characters,
 * once with one malloc'ed NUL-terminated string per pair (pa_batch_align + pa_free_cigars) and once as pointers + lengths into the
 * plan's host buffer (pa_batch_align_view, twice: the second call must leave the first call's costs and texts reproducible).  Costs
 * and texts must agree, every pair must cost what pa_align says.  Prints "batch_view_check ok pairs=<n> chars=<total text length>". */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pa_astarpa2.h"
#include "pa_bitpacking_hip.h"

#define P 48

int main(void) {
    static uint8_t* a[P];
    static uint8_t* b[P];
    static size_t al[P], bl[P];
    unsigned s = 424242;
    for (int i = 0; i < P; ++i) {
        const size_t n = 300 + 45 * (size_t)i;
        a[i] = malloc(n);
        b[i] = malloc(n + 8);
        size_t m = 0;
        for (size_t k = 0; k < n; ++k) {
            s = s * 1103515245u + 12345u;
            a[i][k] = "ACGT"[(s >> 16) & 3];
            s = s * 1103515245u + 12345u;
            const unsigned r = (s >> 16) % 100;
            if (r < 3 + (unsigned)(i % 4) * 3) { /* an edit: substitution, deletion or insertion */
                if (r % 3 == 0) b[i][m++] = a[i][k] == 'A' ? 'C' : 'A';
                else if (r % 3 == 1 && m + 2 < n + 8) { b[i][m++] = a[i][k]; b[i][m++] = 'G'; }
            } else {
                b[i][m++] = a[i][k];
            }
            if (m > n + 6) m = n + 6;
        }
        al[i] = n;
        bl[i] = m;
    }
    size_t chars = 0;
    for (int preset = 0; preset < 2; ++preset) {
        pa_astarpa2_params prm;
        if (preset == 0) pa_params_simple(&prm);
        else pa_params_full(&prm);
        pa_batch* plan = pa_batch_create_params((const uint8_t* const*)a, al, (const uint8_t* const*)b, bl, P, &prm);
        if (!plan) {
            fprintf(stderr, "create: %s\n", pa_last_error());
            return 1;
        }
        int32_t c1[P], c2[P], c3[P];
        char* g1[P];
        const char* t2[P];
        const char* t3[P];
        uint32_t l2[P], l3[P];
        if (pa_batch_align(plan, c1, g1, NULL, NULL) != 0 || pa_batch_align_view(plan, c2, t2, l2, NULL, NULL) != 0) {
            fprintf(stderr, "align: %s\n", pa_last_error());
            return 1;
        }
        for (int i = 0; i < P; ++i) {
            if (c1[i] != c2[i] || strlen(g1[i]) != l2[i] || memcmp(g1[i], t2[i], l2[i]) != 0) {
                fprintf(stderr, "preset %d pair %d: view differs from the string\n", preset, i);
                return 1;
            }
        }
        if (pa_batch_align_view(plan, c3, t3, l3, NULL, NULL) != 0) {
            fprintf(stderr, "align: %s\n", pa_last_error());
            return 1;
        }
        for (int i = 0; i < P; ++i) {
            if (c1[i] != c3[i] || strlen(g1[i]) != l3[i] || memcmp(g1[i], t3[i], l3[i]) != 0) {
                fprintf(stderr, "preset %d pair %d: second view differs\n", preset, i);
                return 1;
            }
            chars += l3[i];
        }
        for (int i = 0; i < P; i += 7) { /* ... and what one call at a time says */
            int32_t cost = -1;
            char* cig = NULL;
            if (pa_align(a[i], al[i], b[i], bl[i], &prm, 1, &cost, &cig, NULL) != 0 || cost != c1[i] || strcmp(cig, g1[i]) != 0) {
                fprintf(stderr, "preset %d pair %d: pa_align disagrees\n", preset, i);
                return 1;
            }
            pa_free_cigars(&cig, 1);
        }
        pa_free_cigars(g1, P);
        pa_batch_destroy(plan);
    }
    printf("batch_view_check ok pairs=%d chars=%zu\n", P, chars);
    return 0;
}
