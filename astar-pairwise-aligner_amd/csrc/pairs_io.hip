// pairs_io.hip -- the data formats either side of the batched aligner (host code only).
//
// What it restates: pa-bin's input loop and result line (pa-bin/src/lib.rs:67-114, pa-bin/src/main.rs:24-35):
//   * `.seq`  : consecutive line pairs, the first starts with '>' and the second with '<' (both markers dropped);
//   * `.txt`  : consecutive line pairs, plain sequences;
//   * `.fna` / `.fa` / `.fasta` : FASTA records taken two at a time (multi-line sequences concatenated);
//   * a directory: every file in it (sorted by name here; the reference uses the directory order);
//   * output: one line `{cost},{cigar}` per pair.
// Lines lose their trailing "\n" / "\r\n" like Rust's BufRead::lines; an odd trailing line or record is dropped like
// itertools' tuples().  Nothing here touches the GPU; pa_align_file feeds the pairs to pa_batch_align.
#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "pa_hip_internal.hpp"

// The sequences are views into the file contents (one read per file, no copy per line: the reader runs at memory speed -- a 200 MB
// .seq file of 10 000 pairs took ten times longer to read line by line than the GPU takes to align it).  FASTA records that span
// several lines are compacted in place.
struct pa_pairs {
    struct View {
        const char* p;
        size_t n;
    };
    struct Blob {  // one file: a private mapping (regular files) or a heap buffer (whatever cannot be mapped)
        char* data = nullptr;
        size_t size = 0;
        bool mapped = false;
    };
    std::vector<Blob> blobs;
    std::vector<View> a, b;
    pa_pairs() = default;
    pa_pairs(const pa_pairs&) = delete;
    pa_pairs& operator=(const pa_pairs&) = delete;
    ~pa_pairs() {
        for (Blob& bl : blobs) {
            if (bl.mapped) munmap(bl.data, bl.size);
            else std::free(bl.data);
        }
    }
};

namespace {

// Whole file -> blob (kept by `out`); false: cannot open / read.  Private and writable: FASTA records are compacted in place
// (copy-on-write touches only those pages), the file itself is never written.
bool slurp(const std::string& path, pa_pairs& out, char** data, size_t* size) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_POPULATE, fd, 0);
        if (m != MAP_FAILED) {
            close(fd);
            out.blobs.push_back({(char*)m, (size_t)st.st_size, true});
            *data = (char*)m;
            *size = (size_t)st.st_size;
            return true;
        }
    }
    size_t cap = 1 << 20, len = 0;
    char* buf = (char*)std::malloc(cap);
    bool ok = buf != nullptr;
    while (ok) {
        if (len == cap) {
            char* bigger = (char*)std::realloc(buf, cap * 2);
            if (!bigger) {
                ok = false;
                break;
            }
            buf = bigger;
            cap *= 2;
        }
        const ssize_t got = read(fd, buf + len, cap - len);
        if (got < 0) ok = false;
        if (got <= 0) break;
        len += (size_t)got;
    }
    close(fd);
    if (!ok) {
        std::free(buf);
        return false;
    }
    out.blobs.push_back({buf, len, false});
    *data = buf;
    *size = len;
    return true;
}

// The lines of a buffer as std::getline yields them (split at '\n', no empty line after a final '\n'), a trailing '\r' dropped.
template <class F>
void for_each_line(char* data, size_t size, F&& f) {
    size_t pos = 0;
    while (pos < size) {
        const char* nl = (const char*)std::memchr(data + pos, '\n', size - pos);
        size_t end = nl ? (size_t)(nl - data) : size;
        size_t len = end - pos;
        if (len && data[pos + len - 1] == '\r') len -= 1;
        f(data + pos, len);
        pos = end + 1;
    }
}

std::string extension(const std::string& path) {
    const size_t slash = path.find_last_of('/');
    const size_t dot = path.find_last_of('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return "";
    return path.substr(dot + 1);
}

int read_file(const std::string& path, pa_pairs& out) {
    const std::string ext = extension(path);
    const bool seq = ext == "seq", txt = ext == "txt", fasta = ext == "fna" || ext == "fa" || ext == "fasta";
    if (!seq && !txt && !fasta) {
        pa::set_error("Unknown file extension \"%s\". Must be in {seq,txt,fna,fa,fasta}.", ext.c_str());
        return PA_E_ARG;
    }
    char* data = nullptr;
    size_t size = 0;
    if (!slurp(path, out, &data, &size)) {
        pa::set_error("cannot open %s", path.c_str());
        return PA_E_ARG;
    }
    if (seq || txt) {
        size_t line_no = 0;
        int rc = 0;
        pa_pairs::View first{nullptr, 0};
        for_each_line(data, size, [&](char* p, size_t n) {
            if (rc != 0) return;
            line_no += 1;
            if (line_no & 1) {  // (an odd last line is dropped like itertools' tuples(): nothing is checked before its partner arrives)
                first = {p, n};
                return;
            }
            pa_pairs::View second{p, n};
            if (seq) {
                if (first.n == 0 || first.p[0] != '>' || second.n == 0 || second.p[0] != '<') {
                    pa::set_error("%s: line %zu: .seq pairs are a '>' line followed by a '<' line", path.c_str(), line_no - 1);
                    rc = PA_E_ARG;
                    return;
                }
                first = {first.p + 1, first.n - 1};
                second = {second.p + 1, second.n - 1};
            }
            out.a.push_back(first);
            out.b.push_back(second);
        });
        return rc;
    }
    // FASTA: records taken two at a time, multi-line sequences concatenated (in place: the write position never passes the read position)
    std::vector<pa_pairs::View> records;
    char* wr = data;
    bool open = false;
    int rc = 0;
    for_each_line(data, size, [&](char* p, size_t n) {
        if (rc != 0) return;
        if (n && p[0] == '>') {
            records.push_back({wr, 0});
            open = true;
        } else if (open) {
            if (wr != p) std::memmove(wr, p, n);
            wr += n;
            records.back().n += n;
        } else if (n) {
            pa::set_error("%s: sequence data before the first FASTA header", path.c_str());
            rc = PA_E_ARG;
        }
    });
    if (rc != 0) return rc;
    for (size_t i = 0; i + 1 < records.size(); i += 2) {
        out.a.push_back(records[i]);
        out.b.push_back(records[i + 1]);
    }
    return 0;
}

}  // namespace

extern "C" pa_pairs* pa_pairs_read(const char* path) {
    if (!path) return nullptr;
    auto p = std::make_unique<pa_pairs>();
    struct stat st;
    if (stat(path, &st) != 0) {
        pa::set_error("%s is not a file or directory", path);
        return nullptr;
    }
    if (S_ISDIR(st.st_mode)) {
        std::vector<std::string> files;
        if (DIR* d = opendir(path)) {
            while (dirent* e = readdir(d)) {
                const std::string name = e->d_name;
                if (name != "." && name != "..") files.push_back(std::string(path) + "/" + name);
            }
            closedir(d);
        }
        std::sort(files.begin(), files.end());
        for (const std::string& f : files)
            if (read_file(f, *p) != 0) return nullptr;
    } else if (read_file(path, *p) != 0) {
        return nullptr;
    }
    return p.release();
}

extern "C" size_t pa_pairs_count(const pa_pairs* p) { return p ? p->a.size() : 0; }

extern "C" int pa_pairs_get(const pa_pairs* p, size_t i, const uint8_t** a, size_t* a_len, const uint8_t** b, size_t* b_len) {
    if (!p || i >= p->a.size()) return PA_E_ARG;
    if (a) *a = reinterpret_cast<const uint8_t*>(p->a[i].p);
    if (a_len) *a_len = p->a[i].n;
    if (b) *b = reinterpret_cast<const uint8_t*>(p->b[i].p);
    if (b_len) *b_len = p->b[i].n;
    return 0;
}

extern "C" void pa_pairs_free(pa_pairs* p) { delete p; }

extern "C" int pa_write_results_csv(const char* path, const int32_t* costs, const char* const* cigars, size_t n) {
    FILE* f = std::fopen(path, "w");
    if (!f) {
        pa::set_error("cannot create %s", path ? path : "(null)");
        return PA_E_ARG;
    }
    for (size_t i = 0; i < n; ++i) std::fprintf(f, "%d,%s\n", costs[i], cigars && cigars[i] ? cigars[i] : "");
    return std::fclose(f) == 0 ? 0 : PA_E_ARG;
}

// pa-bin's main loop for a whole input at once: read the pairs, align them all on the GPU (cost + CIGAR), write the CSV.
extern "C" int pa_align_file(const char* input_path, const char* output_path, size_t* pairs_out) {
    std::unique_ptr<pa_pairs, void (*)(pa_pairs*)> in(pa_pairs_read(input_path), pa_pairs_free);
    if (!in) return PA_E_ARG;
    const size_t n = in->a.size();
    if (pairs_out) *pairs_out = n;
    std::vector<const uint8_t*> ap(n), bp(n);
    std::vector<size_t> al(n), bl(n);
    for (size_t i = 0; i < n; ++i) {
        ap[i] = reinterpret_cast<const uint8_t*>(in->a[i].p);
        bp[i] = reinterpret_cast<const uint8_t*>(in->b[i].p);
        al[i] = in->a[i].n;
        bl[i] = in->b[i].n;
    }
    std::vector<int32_t> costs(n, 0);
    std::vector<char*> cigars(n, nullptr);
    int rc = 0;
    if (n) {
        pa_batch* plan = pa_batch_create_trace(ap.data(), al.data(), bp.data(), bl.data(), n);
        if (!plan) return PA_E_HIP;
        rc = pa_batch_align(plan, costs.data(), cigars.data(), nullptr, nullptr);
        pa_batch_destroy(plan);
    }
    if (rc == 0 && output_path) rc = pa_write_results_csv(output_path, costs.data(), cigars.data(), n);
    for (char* c : cigars) std::free(c);
    return rc;
}

// Many-pair mode over several GPUs from ONE process (the reference aligns pairs one after another,
// pa-bin/src/main.rs:24-35, so they shard with no data-path exchange): a WORK QUEUE.  The pairs are sorted by estimated work
// (heaviest first), cut into chunks, and one host thread per entry of `devices` binds its device (pa_set_device is per thread) and
// pulls chunk after chunk from one atomic counter: pa_batch_create* + pa_batch_align (or the cost-only batch when cigar_out is NULL)
// per chunk, results scattered to the pairs' own indices.  The estimate only orders the queue -- the balance is dynamic, which is
// what band-limited alignment needs (the work of a pair depends on its divergence, which nobody knows beforehand).  A device may
// be listed more than once (two chunks in flight on one GPU: the upload of one overlaps the kernels of the other).
// `params` != NULL: the batched A*PA2 of pa_batch_create_params (the `simple` preset and its relatives).
static int batch_align_queue(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b, const size_t* b_len, size_t pairs,
                             const int* devices, int ndevices, int32_t* cost_out, char** cigar_out, const pa_astarpa2_params* params,
                             pa_astarpa2_stats* stats_out) {
    if (ndevices <= 0 || !devices || !cost_out || (pairs && (!a || !b || !a_len || !b_len)) || (stats_out && !params)) {
        pa::set_error("pa_batch_align_multi: bad arguments");
        return PA_E_ARG;
    }
    const int ndev = pa_device_count();
    for (int d = 0; d < ndevices; ++d)
        if (devices[d] < 0 || devices[d] >= ndev) {
            pa::set_error("pa_batch_align_multi: device %d not visible (%d device(s))", devices[d], ndev);
            return PA_E_ARG;
        }
    if (cigar_out)
        for (size_t i = 0; i < pairs; ++i) cigar_out[i] = nullptr;
    // heaviest first, ties by index: deterministic queue order.  Full DP: n * ceil(m / 64) word updates; band-limited: the band of a
    // pair grows with its length too (n * sqrt-ish), the plain length orders those well enough.
    std::vector<size_t> order(pairs);
    std::vector<uint64_t> work(pairs);
    for (size_t i = 0; i < pairs; ++i) {
        order[i] = i;
        work[i] = params ? (uint64_t)a_len[i] + b_len[i] + 1 : (uint64_t)a_len[i] * ((b_len[i] + 63) / 64) + 1;
    }
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return work[x] != work[y] ? work[x] > work[y] : x < y; });
    // chunks: about eight per device, at least 256 pairs each (a chunk must still fill a GPU), one per device when there are few pairs
    size_t chunk = pairs / ((size_t)ndevices * 8) + 1;
    if (chunk < 256) chunk = 256;
    if (chunk * (size_t)ndevices > pairs) chunk = (pairs + (size_t)ndevices - 1) / (size_t)ndevices;
    if (chunk == 0) chunk = 1;
    if (ndevices == 1) chunk = std::max<size_t>(pairs, 1);  // one worker: nothing to balance, every chunk more is a batch creation more
    if (const char* e = std::getenv("PA_MULTI_CHUNK")) chunk = std::max<size_t>(1, (size_t)std::atoll(e));  // (tests)
    // ... and no chunk's block-column store (traced batches: one V column per 256 columns of a, full height) beyond ~24 GB
    double kChunkBytes = 24e9;
    if (const char* e = std::getenv("PA_MULTI_CHUNK_BYTES")) kChunkBytes = std::max(1.0, std::atof(e));  // (tests)
    auto need_of = [&](size_t i) -> double {
        return (cigar_out || params) ? ((double)a_len[i] / 256.0 + 2.0) * (double)((b_len[i] + 63) / 64) * 16.0 : 0.0;
    };
    std::vector<size_t> bounds{0};
    if (ndevices > 1 && pairs > 0 && chunk * (size_t)ndevices >= pairs && !std::getenv("PA_MULTI_CHUNK")) {
        // Few pairs: one chunk per device, and then the queue cannot correct a bad split -- contiguous slices of the heaviest-first
        // order would hand the first device all the heavy pairs.  Deal them out longest-processing-time-first instead.
        const size_t bins = std::min<size_t>((size_t)ndevices, pairs);
        std::vector<std::vector<size_t>> bin(bins);
        std::vector<uint64_t> load(bins, 0);
        for (size_t k = 0; k < pairs; ++k) {
            const size_t r = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
            bin[r].push_back(order[k]);
            load[r] += work[order[k]];
        }
        // (a bin is still cut where its block-column store would pass the cap: a few hundred long traced pairs dealt into one chunk
        //  per device need far more than a device holds -- 1 GB per 1 Mbp pair)
        size_t pos = 0;
        for (size_t r = 0; r < bins; ++r) {
            double bytes = 0;
            size_t cnt = 0;
            for (size_t i : bin[r]) {
                const double need = need_of(i);
                if (cnt > 0 && bytes + need > kChunkBytes) {
                    bounds.push_back(pos);
                    bytes = 0;
                    cnt = 0;
                }
                order[pos++] = i;
                bytes += need;
                cnt += 1;
            }
            if (cnt > 0) bounds.push_back(pos);
        }
    } else {
        double bytes = 0;
        size_t cnt = 0;
        for (size_t k = 0; k < pairs; ++k) {
            const size_t i = order[k];
            const double need = need_of(i);
            if (cnt > 0 && (cnt >= chunk || bytes + need > kChunkBytes)) {
                bounds.push_back(k);
                bytes = 0;
                cnt = 0;
            }
            bytes += need;
            cnt += 1;
        }
        bounds.push_back(pairs);
    }
    const size_t nchunks = bounds.size() - 1;
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    std::vector<int> rcs((size_t)ndevices, 0);
    std::vector<std::string> errs((size_t)ndevices);
    auto worker = [&](int r) {
        int rc = pa_set_device(devices[r]);
        while (rc == 0 && !failed.load(std::memory_order_relaxed)) {
            const size_t c = next.fetch_add(1, std::memory_order_relaxed);
            if (c >= nchunks) break;
            std::vector<size_t> mine(order.begin() + bounds[c], order.begin() + bounds[c + 1]);
            std::sort(mine.begin(), mine.end());
            const size_t k = mine.size();
            std::vector<const uint8_t*> ap(k), bp(k);
            std::vector<size_t> al(k), bl(k);
            for (size_t j = 0; j < k; ++j) {
                ap[j] = a[mine[j]];
                bp[j] = b[mine[j]];
                al[j] = a_len[mine[j]];
                bl[j] = b_len[mine[j]];
            }
            std::vector<int32_t> costs(k, 0);
            std::vector<char*> cigars(cigar_out ? k : 0, nullptr);
            std::vector<pa_astarpa2_stats> st(stats_out ? k : 0);
            pa_batch* plan = params ? pa_batch_create_params(ap.data(), al.data(), bp.data(), bl.data(), k, params)
                             : cigar_out ? pa_batch_create_trace(ap.data(), al.data(), bp.data(), bl.data(), k)
                                         : pa_batch_create(ap.data(), al.data(), bp.data(), bl.data(), k);
            if (!plan) rc = PA_E_HIP;
            else {
                rc = (cigar_out || params) ? pa_batch_align(plan, costs.data(), cigar_out ? cigars.data() : nullptr, nullptr, nullptr) : pa_batch_run(plan, costs.data(), nullptr);
                if (rc == 0 && stats_out) rc = pa_batch_pair_stats(plan, st.data());
                pa_batch_destroy(plan);
            }
            if (rc == 0)
                for (size_t j = 0; j < k; ++j) {
                    cost_out[mine[j]] = costs[j];
                    if (cigar_out) cigar_out[mine[j]] = cigars[j];
                    if (stats_out) stats_out[mine[j]] = st[j];
                }
            else
                for (char* g : cigars) std::free(g);
        }
        if (rc != 0) {
            failed.store(true, std::memory_order_relaxed);
            rcs[(size_t)r] = rc;
            errs[(size_t)r] = pa_last_error();  // (the error text is per thread)
        }
    };
    std::vector<std::thread> threads;
    for (int r = 1; r < ndevices; ++r) threads.emplace_back(worker, r);
    int cur = 0;
    (void)hipGetDevice(&cur);
    worker(0);  // the calling thread is the first worker
    (void)pa_set_device(cur);
    for (std::thread& t : threads) t.join();
    for (int r = 0; r < ndevices; ++r)
        if (rcs[(size_t)r] != 0) {
            if (cigar_out)
                for (size_t i = 0; i < pairs; ++i) {
                    std::free(cigar_out[i]);
                    cigar_out[i] = nullptr;
                }
            pa::set_error("pa_batch_align_multi: worker %d (device %d): %s", r, devices[r], errs[(size_t)r].c_str());
            return rcs[(size_t)r];
        }
    return 0;
}

// pa-bin's loop with an aligner's parameters (`pa-bin --aligner astarpa2 ...`, pa-bin/src/main.rs:24-35): parameters of the batched
// A*PA2 family run as batches on the current device (chunks of bounded memory), anything else as a loop over pa_align.
extern "C" int pa_align_file_params(const char* input_path, const char* output_path, const pa_astarpa2_params* params, size_t* pairs_out) {
    if (!params) return pa_align_file(input_path, output_path, pairs_out);
    std::unique_ptr<pa_pairs, void (*)(pa_pairs*)> in(pa_pairs_read(input_path), pa_pairs_free);
    if (!in) return PA_E_ARG;
    const size_t n = in->a.size();
    if (pairs_out) *pairs_out = n;
    std::vector<const uint8_t*> ap(n), bp(n);
    std::vector<size_t> al(n), bl(n);
    for (size_t i = 0; i < n; ++i) {
        ap[i] = reinterpret_cast<const uint8_t*>(in->a[i].p);
        bp[i] = reinterpret_cast<const uint8_t*>(in->b[i].p);
        al[i] = in->a[i].n;
        bl[i] = in->b[i].n;
    }
    std::vector<int32_t> costs(n, 0);
    std::vector<char*> cigars(n, nullptr);
    int rc = 0;
    if (n) {
        if (pa_batch_params_supported(params)) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            rc = batch_align_queue(ap.data(), al.data(), bp.data(), bl.data(), n, &dev, 1, costs.data(), cigars.data(), params, nullptr);
        } else {
            for (size_t i = 0; i < n && rc == 0; ++i) rc = pa_align(ap[i], al[i], bp[i], bl[i], params, 1, &costs[i], &cigars[i], nullptr);
        }
    }
    if (rc == 0 && output_path) rc = pa_write_results_csv(output_path, costs.data(), cigars.data(), n);
    for (char* c : cigars) std::free(c);
    return rc;
}

extern "C" int pa_batch_align_multi(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b, const size_t* b_len,
                                    size_t pairs, const int* devices, int ndevices, int32_t* cost_out, char** cigar_out) {
    return batch_align_queue(a, a_len, b, b_len, pairs, devices, ndevices, cost_out, cigar_out, nullptr, nullptr);
}

extern "C" int pa_batch_align_multi_params(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b, const size_t* b_len,
                                           size_t pairs, const int* devices, int ndevices, const pa_astarpa2_params* params, int32_t* cost_out,
                                           char** cigar_out, pa_astarpa2_stats* stats_out) {
    if (!params || !pa_batch_params_supported(params)) {
        pa::set_error("pa_batch_align_multi_params: parameters outside the batched A*PA2 family (see pa_batch_create_params); use pa_align");
        return PA_E_ARG;
    }
    return batch_align_queue(a, a_len, b, b_len, pairs, devices, ndevices, cost_out, cigar_out, params, stats_out);
}
