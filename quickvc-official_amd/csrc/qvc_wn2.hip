// The continuous-stream WaveNet stack kernel (qvc_wn2_impl.h), both operand types: a translation unit of its own so
// that it compiles beside the conv kernels.
#include "qvc_wn2_impl.h"
namespace qvc {
template int launch_wn_stack2_typed<_Float16>(const ConvDesc&, const WnStackArgs&, int, void*);
template int launch_wn_stack2_typed<__bf16>(const ConvDesc&, const WnStackArgs&, int, void*);
bool wn_stack2_supported(const ConvDesc& din, const WnStackArgs& a) { return wn2_supported(din, a); }
}  // namespace qvc
#ifdef QVC_SATCOUNT
namespace qvc { QVC_SAT_READER(sat_count_wn2) }
#endif
