// mapcaller_amd/csrc/mcx_cpus.h — how many CPUs the host side may really use.
// std::thread::hardware_concurrency() counts the machine's; a container is often given a share of them as CPU time (cgroup cpu.max:
// "1600000 100000" = sixteen CPUs' worth per 100 ms on the 256-thread bench box).  Threads beyond that share do not run in parallel: the
// whole group is put to sleep for the rest of each period once the share is spent, the thread that feeds the GPU with it (round 5:
// 60-70 ms stalls in one batch in three at 64 + 64 host threads, none at 12 + 12).  MCX_HOST_CPUS overrides.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <sched.h>

static inline unsigned mcx_usable_cpus()
{
    static const unsigned n = [] {
        if (const char *e = getenv("MCX_HOST_CPUS")) { const int v = atoi(e); if (v > 0) return (unsigned)v; }
        unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) hw = std::min(hw, (unsigned)CPU_COUNT(&set));
        double share = 1e9;
        auto v2 = [&](const std::string &dir) { // cgroup v2: "<quota|max> <period>"
            FILE *f = fopen((dir + "/cpu.max").c_str(), "r");
            if (!f) return;
            char q[64]; double period = 0;
            if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) share = std::min(share, atof(q) / period);
            fclose(f);
        };
        auto v1 = [&](const std::string &dir) { // cgroup v1
            double quota = -1, period = 0;
            if (FILE *f = fopen((dir + "/cpu.cfs_quota_us").c_str(), "r")) { if (fscanf(f, "%lf", &quota) != 1) quota = -1; fclose(f); }
            if (FILE *f = fopen((dir + "/cpu.cfs_period_us").c_str(), "r")) { if (fscanf(f, "%lf", &period) != 1) period = 0; fclose(f); }
            if (quota > 0 && period > 0) share = std::min(share, quota / period);
        };
        // the process's own group and every group above it (the mount is usually the container's: its root carries the container's limit)
        std::string own2, own1;
        if (FILE *f = fopen("/proc/self/cgroup", "r")) {
            char line[4096];
            while (fgets(line, sizeof line, f)) {
                std::string l(line);
                while (!l.empty() && (l.back() == '\n' || l.back() == '\r')) l.pop_back();
                if (l.compare(0, 3, "0::") == 0) own2 = l.substr(3);
                else { const size_t a = l.find(':'), b = a == std::string::npos ? a : l.find(':', a + 1); if (b != std::string::npos && ("," + l.substr(a + 1, b - a - 1) + ",").find(",cpu,") != std::string::npos) own1 = l.substr(b + 1); }
            }
            fclose(f);
        }
        for (std::string p = own2;; p = p.substr(0, p.find_last_of('/'))) { v2("/sys/fs/cgroup" + p); if (p.empty() || p == "/") { v2("/sys/fs/cgroup"); break; } }
        for (std::string p = own1;; p = p.substr(0, p.find_last_of('/'))) { v1("/sys/fs/cgroup/cpu" + p); if (p.empty() || p == "/") { v1("/sys/fs/cgroup/cpu"); break; } }
        if (share < 1e9) hw = std::min(hw, (unsigned)std::max(1.0, share + 0.5));
        return std::max(1u, hw);
    }();
    return n;
}
