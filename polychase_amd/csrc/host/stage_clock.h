// stage_clock.h -- where does the time of a long-running call go?  The tracking / refinement drivers accumulate wall time per
// labelled stage (a handful of scopes per frame: two clock reads each) and, when they finish, file the totals under the
// call's title: POLYCHASE_TRACE_STAGES=1 prints them to stderr, StageClock::Last(title) hands the last report of that title
// to whoever measures (polychase_core._stage_report: bench.py's c5 block, tools/c5_profile.sh).
#pragma once

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <utility>

class StageClock {
   public:
    using Totals = std::map<std::string, std::pair<double, long>>;   // label -> (milliseconds or a count, calls)
    static bool Enabled() {   // printing only; the accumulation is always on
        static const bool on = [] {
            const char* e = std::getenv("POLYCHASE_TRACE_STAGES");
            return e && e[0] && e[0] != '0';
        }();
        return on;
    }
    class Scope {
       public:
        explicit Scope(const char* label) : label_(label), t0_(std::chrono::steady_clock::now()) {}
        ~Scope() {
            auto& e = Running()[label_];
            e.first += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count();
            e.second += 1;
        }

       private:
        const char* label_;
        std::chrono::steady_clock::time_point t0_;
    };
    // a duration measured elsewhere (e.g. by a kernel), or a count, under the same report
    static void Add(const char* label, double value, int calls = 1) {
        auto& e = Running()[label];
        e.first += value;
        e.second += calls;
    }
    // the calling thread's totals become the report `title`; the thread starts from zero again
    static void Report(const char* title) {
        Totals t;
        t.swap(Running());
        if (Enabled()) {
            std::fprintf(stderr, "[polychase stages] %s\n", title);
            for (const auto& kv : t) std::fprintf(stderr, "  %-44s %10.2f ms  %8ld calls\n", kv.first.c_str(), kv.second.first, kv.second.second);
        }
        std::lock_guard<std::mutex> lk(Mutex());
        Reports()[title] = std::move(t);
    }
    // the calling thread starts from zero: the entry of a call that will Report (a call that left without reporting -- stopped by
    // the callback, an exception -- must not leak its totals into the next report of this thread)
    static void Begin() { Running().clear(); }
    static Totals Last(const std::string& title) {
        std::lock_guard<std::mutex> lk(Mutex());
        auto it = Reports().find(title);
        return it == Reports().end() ? Totals{} : it->second;
    }

   private:
    static Totals& Running() {
        static thread_local Totals t;
        return t;
    }
    static std::mutex& Mutex() {
        static std::mutex m;
        return m;
    }
    static std::map<std::string, Totals>& Reports() {
        static std::map<std::string, Totals> r;
        return r;
    }
};
