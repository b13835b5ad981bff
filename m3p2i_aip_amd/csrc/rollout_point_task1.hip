// rollout_point_task1.hip -- the point_env rollout kernel with the reference's default sampler and the task
// `push` compiled in (rollout_point_kernel.hpp); its own translation unit so that the instances build in parallel.
#include "rollout_point_kernel.hpp"

namespace m3 {

void launch_rollout_point_push(const RolloutArgs& a, const PointScene& sc, int blocks, hipStream_t s) {
    launch_rollout_point_instance<false, 1>(a, sc, blocks, s);
}

}  // namespace m3
