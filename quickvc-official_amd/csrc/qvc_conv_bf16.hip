// bf16-operand instantiation of the conv kernel.
#include "qvc_conv_impl.h"
#include "qvc_post_tail_impl.h"
namespace qvc { template int launch_conv_typed<__bf16>(const ConvDesc&, const ConvArgs&, int, int, void*, int*);
template int launch_post_tail_typed<__bf16>(const ConvDesc&, const PostTailArgs&, int, void*);
template int launch_wn_stack_typed<__bf16>(const ConvDesc&, const WnStackArgs&, int, void*);
template int launch_wn_typed<__bf16>(const ConvDesc&, const WnArgs&, int, void*, int*);
template int launch_pair_typed<__bf16, __bf16>(const ConvDesc*, const PairArgs3&, int, void*, int*);
// QVC_BF16X: bf16 operands, f16 residual stream
template int launch_pair_typed<__bf16, _Float16>(const ConvDesc*, const PairArgs3&, int, void*, int*); }
#ifdef QVC_SATCOUNT
namespace qvc { QVC_SAT_READER(sat_count_conv_bf16) }
#endif
