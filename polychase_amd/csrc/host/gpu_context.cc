#include "gpu_context.h"

#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>

pc_context* SharedGpuContext() {
    static std::mutex mtx;
    static pc_context* ctx = nullptr;
    std::lock_guard<std::mutex> lk(mtx);
    if (!ctx) {
        int device = 0;
        if (const char* env = std::getenv("POLYCHASE_DEVICE")) device = std::atoi(env);
        if (pc_context_create(device, &ctx) != PC_OK)
            throw std::runtime_error(std::string("pc_context_create: ") + pc_last_error());
    }
    return ctx;
}

std::recursive_mutex& SharedGpuMutex() {
    static std::recursive_mutex mtx;
    return mtx;
}
