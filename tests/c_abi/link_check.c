/* Links against libastarpa_c_hip.so exactly like a C user of the reference's astarpa-c crate would (same header names,
 * same five symbols; astarpa-c/astarpa.h:15-65).  Prints "<cost> <cigar>" for every entry point on the pair of the
 * reference's own example (cost 2). */
#include <stdio.h>
#include <string.h>

#include "astarpa.h"

int main(void) {
    const char* a = "ACTCGCT";
    const char* b = "AACTCGTT";
    uint8_t* cigar = NULL;
    uintptr_t len = 0;
    uint64_t cost = astarpa2_simple((const uint8_t*)a, strlen(a), (const uint8_t*)b, strlen(b), &cigar, &len);
    printf("astarpa2_simple %llu %s %lu\n", (unsigned long long)cost, (const char*)cigar, (unsigned long)len);
    astarpa_free_cigar(cigar);
    cost = astarpa2_full((const uint8_t*)a, strlen(a), (const uint8_t*)b, strlen(b), &cigar, &len);
    printf("astarpa2_full %llu %s %lu\n", (unsigned long long)cost, (const char*)cigar, (unsigned long)len);
    astarpa_free_cigar(cigar);
    cost = astarpa((const uint8_t*)a, strlen(a), (const uint8_t*)b, strlen(b), &cigar, &len);
    printf("astarpa %llu %s %lu\n", (unsigned long long)cost, (const char*)cigar, (unsigned long)len);
    astarpa_free_cigar(cigar);
    cost = astarpa_gcsh((const uint8_t*)a, strlen(a), (const uint8_t*)b, strlen(b), 1, 15, false, &cigar, &len);
    printf("astarpa_gcsh %llu %s %lu\n", (unsigned long long)cost, (const char*)cigar, (unsigned long)len);
    astarpa_free_cigar(cigar);
    return 0;
}
