// nt_inst_composite.hip -- instantiates the CompositeScene kernels of nt_composite.hpp.  The build compiles this file
// once per dimension (-DNT_INST_N=3 .. 10, in parallel); without the macro every dimension is instantiated here.
// The lean packet kernel up to four dimensions: seven waves a SIMD instead of six.  It has registers to spare (40 VGPRs) and
// waits on scalar loads 38 % of the time, but its 106 SGPRs admit only six 256-thread blocks a CU (MI355X_MICROARCH.md,
// "Residency"): a budget of 96 admits seven -- 120-cell 1.11 -> 1.04 ms a 1080p frame, 600-cell 0.355 -> 0.339 (eight waves
// with 80 SGPRs: 1.09 / 0.36, the spills to VGPR lanes eat the gain).  The attribute is per translation unit because it takes
// no template arguments; the build compiles one unit per dimension.
#if defined(NT_INST_N) && NT_INST_N <= 4 && !defined(NT_PACKET_WAVES4)
#define NT_PACKET_WAVES4 7
#define NT_PACKET_ATTR __attribute__((amdgpu_num_sgpr(96)))
#endif
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
