// tools/ubench_filewrite.cpp — how fast can ONE file in tmpfs take text from several threads?  (the SAM file of the CLI)
//   g++ -O2 -std=c++17 tools/ubench_filewrite.cpp -o tools/ubench_filewrite -lpthread
//   tools/ubench_filewrite <dir> [GiB] : pwrite from 1..32 threads at disjoint offsets, the same after fallocate, and memcpy into a mapping
#include <atomic>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm";
    const size_t total = (size_t)(argc > 2 ? atof(argv[2]) : 2.0) << 30, piece = (size_t)8 << 20;
    std::vector<char> src((size_t)1 << 30, 'A'); // (a gigabyte of source text: not a piece that stays in the caches)
    const size_t n_src = src.size() / piece;
    const std::string path = dir + "/ubench_filewrite.tmp";
    for (int mode = 0; mode < 5; mode++) {
        for (int T : {1, 2, 4, 8, 12, 16}) {
            unlink(path.c_str());
            const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0600);
            if (fd < 0) { perror("open"); return 1; }
            char *map = nullptr;
            const double tf = now();
            if ((mode == 1 || mode == 2) && posix_fallocate(fd, 0, (off_t)total) != 0) { perror("fallocate"); return 1; }
            if ((mode == 1 || mode == 2) && T == 1) printf("posix_fallocate of the file: %.2f GB/s\n", total / (now() - tf) / 1e9);
            if (mode >= 3 && ftruncate(fd, (off_t)total) != 0) { perror("ftruncate"); return 1; }
            if (mode >= 2) { map = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); if (map == MAP_FAILED) { perror("mmap"); return 1; } }
            const size_t n_piece = total / piece;
            const double t0 = now();
            std::vector<std::thread> pool;
            std::atomic<size_t> ready(mode == 4 ? 0 : total);
            std::thread ahead;
            if (mode == 4) ahead = std::thread([&] { for (size_t at = 0; at < total; at += (size_t)64 << 20) { if (fallocate(fd, 0, (off_t)at, (off_t)std::min<size_t>((size_t)64 << 20, total - at)) != 0) perror("fallocate"); ready.store(at + ((size_t)64 << 20)); } });
            for (int t = 0; t < T; t++) pool.emplace_back([&, t] {
                for (size_t k = t; k < n_piece; k += T) {
                    while (ready.load() < (k + 1) * piece) std::this_thread::yield();
                    if (mode >= 2) memcpy(map + k * piece, src.data() + (k % n_src) * piece, piece);
                    else if (pwrite(fd, src.data() + (k % n_src) * piece, piece, (off_t)(k * piece)) != (ssize_t)piece) { perror("pwrite"); exit(1); }
                }
            });
            for (auto &th : pool) th.join();
            if (ahead.joinable()) ahead.join();
            const double dt = now() - t0;
            printf("%-28s %2d threads  %6.2f GB/s\n", mode == 0 ? "pwrite, growing file" : (mode == 1 ? "pwrite after fallocate" : (mode == 2 ? "memcpy into mmap (fallocated)" : (mode == 3 ? "memcpy into mmap (sparse)" : "mmap, fallocate running ahead"))), T, total / dt / 1e9);
            if (map) munmap(map, total);
            close(fd);
        }
    }
    unlink(path.c_str());
    return 0;
}
