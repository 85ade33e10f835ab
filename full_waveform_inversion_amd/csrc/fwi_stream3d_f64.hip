// Instantiations of the 3-D stream kernel (fwi_stream3d.h): fp64 (double2 lanes), O(2) / O(4) / O(8).
#include "fwi_stream3d.h"

namespace fwi {

template hipError_t launch_stream_r<double, 1>(const GridDesc &, const StepArgs<double> &, const StreamTuning &, hipStream_t);
template hipError_t launch_stream_r<double, 2>(const GridDesc &, const StepArgs<double> &, const StreamTuning &, hipStream_t);
template hipError_t launch_stream_r<double, 4>(const GridDesc &, const StepArgs<double> &, const StreamTuning &, hipStream_t);

}  // namespace fwi
