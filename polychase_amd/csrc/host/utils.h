// utils.h -- CHECK macros with the reference's failure behaviour: std::logic_error with
// "[file:line func] Assertion failed: expr -- message" (reference cpp/utils.h:12-45, cpp/utils.cc:8-18).
#pragma once

#include <cstdarg>
#include <cstdio>
#include <sstream>
#include <stdexcept>
#include <string>

namespace asserts {
[[noreturn]] inline void assert_fail(const char* assertion, const char* file, unsigned line, const char* function,
                                     const std::string& message = std::string()) {
    std::ostringstream os;
    os << "[" << file << ":" << line << " " << function << "] Assertion failed: " << assertion;
    if (!message.empty()) os << " -- " << message;
    throw std::logic_error(os.str());
}
}  // namespace asserts

#define CHECK(expr)                                                                  \
    do {                                                                             \
        if (!static_cast<bool>(expr)) ::asserts::assert_fail(#expr, __FILE__, __LINE__, __func__); \
    } while (0)

#define CHECK_OP(e1, op, e2)                                                                       \
    do {                                                                                           \
        auto&& v1__ = (e1);                                                                        \
        auto&& v2__ = (e2);                                                                        \
        if (!static_cast<bool>(v1__ op v2__)) {                                                    \
            std::ostringstream os__;                                                               \
            os__ << "(" #e1 ") = " << v1__ << " while (" #e2 ") = " << v2__;                       \
            ::asserts::assert_fail("(" #e1 ") " #op " (" #e2 ")", __FILE__, __LINE__, __func__, os__.str()); \
        }                                                                                          \
    } while (0)

#define CHECK_EQ(a, b) CHECK_OP(a, ==, b)
#define CHECK_NE(a, b) CHECK_OP(a, !=, b)
#define CHECK_GT(a, b) CHECK_OP(a, >, b)
#define CHECK_LT(a, b) CHECK_OP(a, <, b)
#define CHECK_GE(a, b) CHECK_OP(a, >=, b)
#define CHECK_LE(a, b) CHECK_OP(a, <=, b)

inline std::string StrFormat(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
inline std::string StrFormat(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return std::string(buf);
}
