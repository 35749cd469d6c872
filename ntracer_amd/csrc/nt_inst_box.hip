// nt_inst_box.hip -- instantiates the BoxScene kernels of nt_box.hpp.  The build compiles this file once per dimension
// (-DNT_INST_N=3 .. 24, in parallel); without the macro every dimension is instantiated here.
// Dimensions whose tile kernel sits a few registers above an occupancy step are held to the step (`amdgpu_waves_per_eu` takes no
// template arguments, hence per translation unit; -DNT_TILE_OCC=... overrides).  Measured (tools/il_ab.py, tools/boxn_time.py,
// settled clocks): n = 10, 100 VGPRs -> 95, five waves a SIMD instead of four: sixteen 4096 x 4096 frames 337 -> 309 us (-8 %);
// n = 15, 132 -> 128, four instead of three: 414 -> 439 Grays/s; n = 21..24, 178..199 -> 168 (a few dwords a lane spilled),
// three instead of two: 204 -> 270, 196 -> 260, 193 -> 249, 192 -> 223 Grays/s.  No gain, not taken: n = 6 at seven waves
// (72 VGPRs, three dwords spilled: 353.9 vs 353.6 us on the headline), n = 7, 8 at six, n = 16 at four.
#if defined(NT_INST_N) && !defined(NT_TILE_OCC)
#if NT_INST_N <= 5
// (up to five dimensions the tile kernel needs 55..70 VGPRs -- seven or eight waves a SIMD -- but its 106 SGPRs admit six:
// ⌊800 / (⌈sgpr/16⌉·16 + 16)⌋, MI355X_MICROARCH.md "Residency".  With a budget of 96 (94 used, nothing spilled) seven are
// admitted: BoxScene(3) 1080p, 160 frames 292.7 -> 285.5 us a call; 80 -- eight waves -- measures the same)
#define NT_TILE_OCC __attribute__((amdgpu_num_sgpr(96)))
#elif NT_INST_N == 6
// (seven waves a SIMD instead of six: 71 VGPRs once up[1..5] are no longer pinned in vector registers, nothing spilled, and the SGPR
// budget that lets the hardware admit the seventh: headline call 350.8 -> 345.2 us)
#define NT_TILE_OCC __attribute__((amdgpu_waves_per_eu(7, 7), amdgpu_num_sgpr(96)))
#define NT_BOX_PIN_UP 0
#elif NT_INST_N == 10
#define NT_TILE_OCC __attribute__((amdgpu_waves_per_eu(5, 5)))
#elif NT_INST_N == 15
#define NT_TILE_OCC __attribute__((amdgpu_waves_per_eu(4, 4)))
#elif NT_INST_N >= 21 && NT_INST_N <= 24
#define NT_TILE_OCC __attribute__((amdgpu_waves_per_eu(3, 3)))
#endif
#endif
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
