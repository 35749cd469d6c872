// nt_launch.cpp -- the thread-local message of the last failed kernel launch, shared by the kernel translation units.
#include "nt_device.hpp"

namespace {
thread_local char g_launch_error[256] = "";
}

char *nt_launch_error_buf() { return g_launch_error; }
const char *nt_launch_error() { return g_launch_error; }
