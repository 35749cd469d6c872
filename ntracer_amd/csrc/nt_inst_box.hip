// nt_inst_box.hip -- instantiates the BoxScene kernels of nt_box.hpp.  The build compiles this file once per dimension
// (-DNT_INST_N=3 .. 24, in parallel); without the macro every dimension is instantiated here.
#include "nt_box.hpp"

#define NT_DEFINE_BOX(N) \
    int nt_box_fixed_##N(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg) { return launch_box_fixed<N>(li, cam, tg); }
#define NT_DEFINE_BOX_(N) NT_DEFINE_BOX(N)

#ifdef NT_INST_N
NT_DEFINE_BOX_(NT_INST_N)
#else
NT_DEFINE_BOX(3) NT_DEFINE_BOX(4) NT_DEFINE_BOX(5) NT_DEFINE_BOX(6) NT_DEFINE_BOX(7) NT_DEFINE_BOX(8) NT_DEFINE_BOX(9) NT_DEFINE_BOX(10)
NT_DEFINE_BOX(11) NT_DEFINE_BOX(12) NT_DEFINE_BOX(13) NT_DEFINE_BOX(14) NT_DEFINE_BOX(15) NT_DEFINE_BOX(16)
NT_DEFINE_BOX(17) NT_DEFINE_BOX(18) NT_DEFINE_BOX(19) NT_DEFINE_BOX(20) NT_DEFINE_BOX(21) NT_DEFINE_BOX(22) NT_DEFINE_BOX(23) NT_DEFINE_BOX(24)
#endif
