// kernel_stubs.cpp -- SANITIZER BUILDS OF THE HOST CODE ONLY (tools/sanitize.sh).  The host side of libntracer_hip.so
// (nt_api.cpp, nt_builder.cpp, nt_launch.cpp) is compiled by g++ with -fsanitize=...; the kernels are not (GPU sanitizers are
// not available on this pool), so the launchers they would provide fail loudly here.  Nothing in the product links this file.
#include <cstdio>
#include "../../ntracer_amd/csrc/nt_device.hpp"

char *nt_launch_error_buf();
static int no_kernels() {
    snprintf(nt_launch_error_buf(), 256, "sanitizer build of the host code: no kernels in this library");
    return -1;
}
int nt_launch_box(const NtLaunchInfo &, const NtCamera &, const NtTarget &) { return no_kernels(); }
int nt_launch_composite(const NtLaunchInfo &, const NtCamera &, const NtCompositeDev &, const NtTarget &) { return no_kernels(); }
int nt_launch_upload(void *, const float *, float *, int) { return no_kernels(); }
int nt_var_frame_words(int n) { return 4 * n + 16; }
