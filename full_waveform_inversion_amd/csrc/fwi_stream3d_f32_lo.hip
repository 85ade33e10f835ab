// Instantiations of the 3-D stream kernel (fwi_stream3d.h): fp32, O(2) and O(4).
#include "fwi_stream3d.h"

namespace fwi {

template hipError_t launch_stream_r<float, 1>(const GridDesc &, const StepArgs<float> &, const StreamTuning &, hipStream_t);
template hipError_t launch_stream_r<float, 2>(const GridDesc &, const StepArgs<float> &, const StreamTuning &, hipStream_t);

}  // namespace fwi
