// iiv_team.hip -- the stand-alone kernel around team_body (iiv_team.h): the greedy selection loop for
// few streams, eight waves per stream.
#include "iiv_team.h"

namespace iiv {

template <int MODE, bool FOUR>
__global__ __launch_bounds__(64 * kScoringWaves) void greedy_team_kernel(StreamState *__restrict__ states,
                                                                         const uint8_t *__restrict__ frames_main,
                                                                         const uint8_t *__restrict__ frames_aux, int n_frames,
                                                                         const LaunchSeg *__restrict__ segs, int seg_stride,
                                                                         const NarrowTables nt,
                                                                         uint8_t *__restrict__ ops_out, size_t ops_stride,
                                                                         unsigned long long *__restrict__ live, uint32_t live_tag)
{
    const LaunchSeg g = segs[(size_t)blockIdx.x * seg_stride];
    team_body<MODE, kScoringWaves, FOUR>(states, frames_main, frames_aux, n_frames, g, nt, ops_out, ops_stride, live, live_tag);
}

int launch_greedy_team(int mode, const GreedyArgs &a, hipStream_t st)
{
#define IIV_TEAM(M, F)                                                                                                        \
    hipLaunchKernelGGL((greedy_team_kernel<M, F>), dim3(a.n_streams), dim3(64 * kScoringWaves), 0, st, a.states, a.frames_main, \
                       a.frames_aux, a.n_frames, a.segs, a.seg_stride, a.nt, a.ops_out, a.ops_stride, a.live, a.live_tag)
    if (mode == kDHGR) {
        if (a.fourth) IIV_TEAM(kDHGR, true);
        else IIV_TEAM(kDHGR, false);
    } else {
        if (a.fourth) IIV_TEAM(kHGR, true);
        else IIV_TEAM(kHGR, false);
    }
#undef IIV_TEAM
    return hip_check(hipGetLastError(), "greedy_team_kernel launch");
}

}  // namespace iiv
