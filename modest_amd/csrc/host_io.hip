// Host-side file ingest of the PP stage: the `.bin` frames of a group of scans read back to back into one staging buffer.
//
// The reference reads one frame per call (np.fromfile in load_velo_scan, utils/pointcloud_utils.py:22-25, called per history frame
// from pre_compute_pp_score.py:137-146).  Here the files of up to four scans (44 new frames, 361 for a cold scan) are stat'ed and
// read by a few host threads inside ONE call that the Python side makes with the interpreter lock released: reader threads in the
// interpreter took the lock three times per file (open / readinto / close) and waited for the scan loop's thread each time --
// a worker's reads ran 1.5-1.7 x slower whenever they overlapped the loop (profiles/r06_cli_slots.txt against r06_cli_slots_creader.txt).
#include "../../include/modest_hip.h"

#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <exception>
#include <fcntl.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

extern "C" int64_t modest_host_read_files(const char *const *paths, int n_files, void *dst, uint64_t capacity_bytes,
                                          uint64_t *sizes_out, int n_threads) {
    if (n_files < 0 || (n_files > 0 && (paths == nullptr || sizes_out == nullptr))) return -1;
    if (n_files == 0) return 0;
    const int nt = n_threads < 1 ? 1 : (n_threads > n_files ? n_files : n_threads);
    std::vector<int> fds((size_t)n_files, -1);
    std::atomic<int> next{0}, bad{n_files};   // bad: smallest index of a file that failed
    auto fail = [&](int k) {
        int cur = bad.load();
        while (k < cur && !bad.compare_exchange_weak(cur, k)) {
        }
    };
    auto run = [&](auto &&body) {
        next.store(0);
        std::vector<std::thread> th;
        try {
            for (int t = 1; t < nt; ++t) th.emplace_back(body);
        } catch (const std::exception &) {   // (no more threads to be had: the ones that started and this one share the files)
        }
        body();
        for (auto &x : th) x.join();
    };
    // pass 1: open + size of every file (the offsets are the running sum: files lie back to back)
    run([&] {
        for (int k; (k = next.fetch_add(1)) < n_files;) {
            struct stat st;
            const int fd = open(paths[k], O_RDONLY | O_CLOEXEC);
            if (fd < 0 || fstat(fd, &st) != 0) {
                if (fd >= 0) close(fd);
                fail(k);
                continue;
            }
            fds[(size_t)k] = fd;
            sizes_out[k] = (uint64_t)st.st_size;
        }
    });
    auto close_all = [&] {
        for (int fd : fds)
            if (fd >= 0) close(fd);
    };
    if (bad.load() < n_files) {
        close_all();
        return -(int64_t)(bad.load() + 2);   // -(k + 2): file k could not be opened
    }
    std::vector<uint64_t> at((size_t)n_files + 1, 0);
    for (int k = 0; k < n_files; ++k) at[(size_t)k + 1] = at[(size_t)k] + sizes_out[k];
    if (at[(size_t)n_files] > capacity_bytes || dst == nullptr) {   // the caller grows its staging buffer and calls again
        close_all();
        return (int64_t)at[(size_t)n_files];
    }
    // pass 2: the reads (a file is one work item: frames are ~0.5 MB, a few dozen per call)
    run([&] {
        for (int k; (k = next.fetch_add(1)) < n_files;) {
            char *p = static_cast<char *>(dst) + at[(size_t)k];
            uint64_t left = sizes_out[k], off = 0;
            while (left > 0) {
                const ssize_t got = pread(fds[(size_t)k], p + off, (size_t)left, (off_t)off);
                if (got < 0 && errno == EINTR) continue;
                if (got <= 0) {   // error, or the file shrank under us
                    fail(k);
                    break;
                }
                off += (uint64_t)got;
                left -= (uint64_t)got;
            }
        }
    });
    close_all();
    if (bad.load() < n_files) return -(int64_t)(bad.load() + 2);
    return 0;
}
