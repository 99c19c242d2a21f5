// qvc_pack_util.h -- host helper shared by the packer and the unit-test conv entry point.
#pragma once
#include "qvc_plan.h"
namespace qvc {
// Packs a plain Conv1d weight [Cout][Cin][k] (+ optional bias) into `base` at d.w_off / d.b_off.
void pack_plain_conv(const ConvDesc& d, const float* w, const float* bias, int dtype, char* base);
}
