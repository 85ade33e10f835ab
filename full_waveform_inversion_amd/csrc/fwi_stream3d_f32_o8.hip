// Instantiations of the 3-D stream kernel (fwi_stream3d.h): fp32, O(8): the headline kernel and every CPML variant of it.
#include "fwi_stream3d.h"

namespace fwi {

template hipError_t launch_stream_r<float, 4>(const GridDesc &, const StepArgs<float> &, const StreamTuning &, hipStream_t);

}  // namespace fwi
