/*
 * oracle/combine_emu_tsan.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Driver of oracle/combine_emu.cpp under ThreadSanitizer: `make -C oracle tsan_combine && oracle/_build/combine_emu_tsan 24 40`.
 */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
extern "C" int pa_combine_emu_run(int threads, int calls, int batch_us, int fail_every, int64_t* out);
int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 24, calls = argc > 2 ? atoi(argv[2]) : 40;
    int64_t a[7], b[7];
    pa_combine_emu_run(threads, calls, 300, 0, a);
    pa_combine_emu_run(threads, calls, 300, 5, b);
    printf("threads %d calls %d: wrong %lld batches %lld largest group %lld side by side %lld; with failing batches: wrong %lld failed requests %lld\n", threads, calls,
           (long long)a[0], (long long)a[1], (long long)a[2], (long long)a[3], (long long)b[0], (long long)b[4]);
    return (a[0] || b[0] || a[4] || a[5] || b[5] || a[1] >= (int64_t)threads * calls) ? 1 : 0;  // (a[5], b[5]: empty groups that were run)
}
