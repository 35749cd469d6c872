// nt_inst_composite.hip -- instantiates the CompositeScene kernels of nt_composite.hpp.  The build compiles this file
// once per dimension (-DNT_INST_N=3 .. 10, in parallel); without the macro every dimension is instantiated here.
#include "nt_composite.hpp"

#define NT_DEFINE_COMPOSITE(N)                                                                                            \
    int nt_composite_fixed_##N(const NtLaunchInfo &li, const NtCamera &cam, const NtCompositeDev &sc, const NtTarget &tg) { \
        return launch_composite_fixed<N>(li, cam, sc, tg);                                                                \
    }
#define NT_DEFINE_COMPOSITE_(N) NT_DEFINE_COMPOSITE(N)

#ifdef NT_INST_N
NT_DEFINE_COMPOSITE_(NT_INST_N)
#else
NT_DEFINE_COMPOSITE(3) NT_DEFINE_COMPOSITE(4) NT_DEFINE_COMPOSITE(5) NT_DEFINE_COMPOSITE(6)
NT_DEFINE_COMPOSITE(7) NT_DEFINE_COMPOSITE(8) NT_DEFINE_COMPOSITE(9) NT_DEFINE_COMPOSITE(10)
#endif
