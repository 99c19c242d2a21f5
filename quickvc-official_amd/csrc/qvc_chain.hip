// The chained-ResBlock kernel (qvc_chain_impl.h) for the three operand / stream type combinations: a translation unit
// of its own so that it compiles beside the conv kernels.
#include "qvc_chain_impl.h"
namespace qvc {
template int launch_chain_typed<_Float16, _Float16>(const ConvDesc*, const ConvDesc*, ChainArgs, int, void*, int*);
template int launch_chain_typed<__bf16, __bf16>(const ConvDesc*, const ConvDesc*, ChainArgs, int, void*, int*);
template int launch_chain_typed<__bf16, _Float16>(const ConvDesc*, const ConvDesc*, ChainArgs, int, void*, int*);
}  // namespace qvc
#ifdef QVC_SATCOUNT
namespace qvc { QVC_SAT_READER(sat_count_chain) }
#endif
