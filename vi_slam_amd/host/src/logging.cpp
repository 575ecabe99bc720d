#include "common/logging.hpp"
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace vi_slam {

LogLevel& logLevel() {
    static LogLevel lvl = [] {
        const char* e = std::getenv("VI_SLAM_LOG");
        if (!e) return LogLevel::WARN;
        if (!std::strcmp(e, "debug")) return LogLevel::DEBUG;
        if (!std::strcmp(e, "info")) return LogLevel::INFO;
        if (!std::strcmp(e, "error")) return LogLevel::ERROR;
        if (!std::strcmp(e, "off")) return LogLevel::OFF;
        return LogLevel::WARN;
    }();
    return lvl;
}

void logWrite(LogLevel lvl, const char* module, const char* fmt, ...) {
    if (lvl < logLevel()) return;
    static std::mutex mu;
    static const char* names[] = {"DEBUG", "INFO", "WARN", "ERROR"};
    std::lock_guard<std::mutex> g(mu);
    std::fprintf(stderr, "[%s] [%s] ", names[(int)lvl], module);
    va_list ap;
    va_start(ap, fmt);
    std::vfprintf(stderr, fmt, ap);
    va_end(ap);
    std::fputc('\n', stderr);
}

}  // namespace vi_slam
