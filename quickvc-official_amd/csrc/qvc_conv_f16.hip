// fp16-operand instantiation of the conv kernel (own translation unit so the two operand
// types compile in parallel).
#include "qvc_conv_impl.h"
#include "qvc_post_tail_impl.h"
namespace qvc { template int launch_conv_typed<_Float16>(const ConvDesc&, const ConvArgs&, int, int, void*, int*);
template int launch_post_tail_typed<_Float16>(const ConvDesc&, const PostTailArgs&, int, void*);
template int launch_wn_stack_typed<_Float16>(const ConvDesc&, const WnStackArgs&, int, void*);
template int launch_wn_typed<_Float16>(const ConvDesc&, const WnArgs&, int, void*, int*);
template int launch_pair_typed<_Float16, _Float16>(const ConvDesc*, const PairArgs3&, int, void*, int*); }
#ifdef QVC_SATCOUNT
namespace qvc { QVC_SAT_READER(sat_count_conv_f16) }
#endif
