#include "numa_pin.h"

#include <sched.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <sstream>
#include <vector>

#include "../../../include/polychase_hip.h"

namespace numa {
namespace {

bool Enabled() {
    const char* e = std::getenv("POLYCHASE_NUMA_PIN");
    return !(e && e[0] == '0');
}

std::string ReadLine(const std::string& path) {
    std::string out;
    if (FILE* f = std::fopen(path.c_str(), "r")) {
        char buf[4096];
        if (std::fgets(buf, sizeof buf, f)) out = buf;
        std::fclose(f);
    }
    while (!out.empty() && (out.back() == '\n' || out.back() == ' ')) out.pop_back();
    return out;
}

// "0-63,128-191" -> cpu set
bool ParseCpuList(const std::string& list, cpu_set_t* set) {
    CPU_ZERO(set);
    bool any = false;
    const char* p = list.c_str();
    while (*p) {
        char* end = nullptr;
        const long a = std::strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = std::strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) {
            CPU_SET(static_cast<int>(c), set);
            any = true;
        }
        if (*p == ',') p++;
    }
    return any;
}

std::string CpuSetToList(const cpu_set_t& set) {
    std::ostringstream os;
    int start = -1;
    bool first = true;
    for (int c = 0; c <= CPU_SETSIZE; c++) {
        const bool on = c < CPU_SETSIZE && CPU_ISSET(c, &set);
        if (on && start < 0) start = c;
        if (!on && start >= 0) {
            if (!first) os << ",";
            first = false;
            if (c - 1 == start) os << start;
            else os << start << "-" << c - 1;
            start = -1;
        }
    }
    return os.str();
}

struct State {
    std::mutex m;
    bool looked = false;
    int node = -1;
    std::string bus_id, node_cpus;
    cpu_set_t node_set;
    std::map<std::string, std::pair<std::string, std::string>> roles;   // role -> (allowed before, allowed after)
};
State& S() {
    static State* s = new State;   // never destroyed: threads may report at exit
    return *s;
}

void Look(pc_context* ctx) {
    State& s = S();
    if (s.looked || !ctx) return;   // (threads that have no context of their own pass null: they use what a caller found)
    s.looked = true;
    char bus[64] = {0};
    if (pc_context_pci_bus_id(ctx, bus, sizeof bus) != PC_OK) return;
    s.bus_id = bus;
    for (char& ch : s.bus_id) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
    const std::string node = ReadLine("/sys/bus/pci/devices/" + s.bus_id + "/numa_node");
    if (node.empty()) return;
    s.node = std::atoi(node.c_str());
    if (s.node < 0) return;
    s.node_cpus = ReadLine("/sys/devices/system/node/node" + std::to_string(s.node) + "/cpulist");
    if (!ParseCpuList(s.node_cpus, &s.node_set)) s.node = -1;
}

}  // namespace

int GpuNode(pc_context* ctx) {
    State& s = S();
    std::lock_guard<std::mutex> lk(s.m);
    Look(ctx);
    return s.node;
}

bool PinThisThreadNearGpu(pc_context* ctx, const char* role) {
    if (!Enabled()) return false;
    State& s = S();
    std::lock_guard<std::mutex> lk(s.m);
    Look(ctx);
    cpu_set_t cur;
    CPU_ZERO(&cur);
    if (sched_getaffinity(0, sizeof cur, &cur) != 0) return false;
    auto& entry = s.roles[role ? role : "thread"];
    entry.first = CpuSetToList(cur);
    entry.second = entry.first;
    if (s.node < 0) return false;
    cpu_set_t want;
    CPU_AND(&want, &cur, &s.node_set);
    if (CPU_COUNT(&want) == 0 || CPU_EQUAL(&want, &cur)) return false;   // the cpuset excludes the node / nothing to narrow
    if (sched_setaffinity(0, sizeof want, &want) != 0) return false;
    entry.second = CpuSetToList(want);
    return true;
}

ScopedPin::ScopedPin(pc_context* ctx, const char* role) {
    static_assert(sizeof(cpu_set_t) <= sizeof(saved_), "cpu_set_t fits");
    cpu_set_t cur;
    CPU_ZERO(&cur);
    if (sched_getaffinity(0, sizeof cur, &cur) != 0) return;
    std::memcpy(saved_, &cur, sizeof cur);
    changed_ = PinThisThreadNearGpu(ctx, role);
}
ScopedPin::~ScopedPin() {
    if (!changed_) return;
    cpu_set_t old;
    std::memcpy(&old, saved_, sizeof old);
    (void)sched_setaffinity(0, sizeof old, &old);
}

std::string Placement() {
    State& s = S();
    std::lock_guard<std::mutex> lk(s.m);
    std::ostringstream os;
    int nodes = 0;
    for (; nodes < 64; nodes++)
        if (ReadLine("/sys/devices/system/node/node" + std::to_string(nodes) + "/cpulist").empty()) break;
    os << "{\"enabled\": " << (Enabled() ? "true" : "false") << ", \"numa_nodes\": " << nodes << ", \"gpu_pci_bus_id\": \"" << s.bus_id
       << "\", \"gpu_numa_node\": " << s.node << ", \"gpu_node_cpus\": \"" << s.node_cpus << "\", \"threads\": {";
    bool first = true;
    for (const auto& kv : s.roles) {
        if (!first) os << ", ";
        first = false;
        os << "\"" << kv.first << "\": {\"allowed_before\": \"" << kv.second.first << "\", \"allowed_after\": \"" << kv.second.second << "\"}";
    }
    os << "}}";
    return os.str();
}

}  // namespace numa
