// stage_clock.h -- where does the host time of a long-running call go?  POLYCHASE_TRACE_STAGES=1 makes the
// tracking / refinement drivers print accumulated wall time per labelled stage to stderr when they finish.
#pragma once

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>

class StageClock {
   public:
    static bool Enabled() {
        static const bool on = [] {
            const char* e = std::getenv("POLYCHASE_TRACE_STAGES");
            return e && e[0] && e[0] != '0';
        }();
        return on;
    }
    class Scope {
       public:
        explicit Scope(const char* label) : label_(Enabled() ? label : nullptr) {
            if (label_) t0_ = std::chrono::steady_clock::now();
        }
        ~Scope() {
            if (!label_) return;
            auto& e = Totals()[label_];
            e.first += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count();
            e.second += 1;
        }

       private:
        const char* label_;
        std::chrono::steady_clock::time_point t0_;
    };
    // a duration measured elsewhere (e.g. by a kernel) under the same report
    static void Add(const char* label, double ms) {
        if (!Enabled()) return;
        auto& e = Totals()[label];
        e.first += ms;
        e.second += 1;
    }
    static void Report(const char* title) {
        if (!Enabled()) return;
        std::fprintf(stderr, "[polychase stages] %s\n", title);
        for (const auto& kv : Totals())
            std::fprintf(stderr, "  %-44s %10.2f ms  %8ld calls\n", kv.first.c_str(), kv.second.first, kv.second.second);
        Totals().clear();
    }

   private:
    static std::map<std::string, std::pair<double, long>>& Totals() {
        static thread_local std::map<std::string, std::pair<double, long>> t;
        return t;
    }
};
