#ifndef VI_SLAM_AMD_LOGGING_HPP
#define VI_SLAM_AMD_LOGGING_HPP
/* Minimal stderr logger with the reference's macro names (include/common/logging.hpp:207-266);
 * printf-style formats.  LOG_DEBUG is compiled out under NDEBUG, as in the reference. */
#include <cstdio>

namespace vi_slam {
enum class LogLevel { DEBUG = 0, INFO = 1, WARN = 2, ERROR = 3, OFF = 4 };
LogLevel& logLevel();
void logWrite(LogLevel lvl, const char* module, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
}  // namespace vi_slam

#ifdef NDEBUG
#define LOG_DEBUG(module, ...) ((void)0)
#else
#define LOG_DEBUG(module, ...) ::vi_slam::logWrite(::vi_slam::LogLevel::DEBUG, module, __VA_ARGS__)
#endif
#define LOG_INFO(module, ...) ::vi_slam::logWrite(::vi_slam::LogLevel::INFO, module, __VA_ARGS__)
#define LOG_WARN(module, ...) ::vi_slam::logWrite(::vi_slam::LogLevel::WARN, module, __VA_ARGS__)
#define LOG_ERROR(module, ...) ::vi_slam::logWrite(::vi_slam::LogLevel::ERROR, module, __VA_ARGS__)
#endif
