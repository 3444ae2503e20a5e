// block.hip -- one C call per torchvision BasicBlock (stride 1, no shortcut convolution) of the ResNet trunks (SURVEY 8f-1 / 8f-3:
// model/graph.py:50-54 encoder, model/view_estimator.py:40-42 estimator):
//     out = relu( bn2( conv2( relu( bn1( conv1(x) ) ) ) ) + x )
// Host-side glue only: the launches are those of sc_conv3x3_forward[_split], sc_bn_act_forward / _backward and sc_conv3x3_wgrad[_split], in
// the order functional.BasicBlockFunction issued them one Python wrapper at a time (so values and gradients are bit-identical); what goes
// away is per launch ~10 us of interpreter / ctypes work -- 18 such blocks per step, 4 launches forward and 7-8 backward each, were 3 ms of
// the ~15 ms the host spends on a training step (the bs16 configuration is host-paced).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "shapeclipper_hip.h"


extern "C" int sc_basic_block_forward(const sc_block_args* a, void* stream) {
    if (!a || a->batch <= 0 || a->channels <= 0) return (int)hipErrorInvalidValue;
    const int B = a->batch, C = a->channels, hw = a->hw, HW = hw * hw, G = a->groups;
    auto conv = a->split ? sc_conv3x3_forward_split : sc_conv3x3_forward;
    int rc = conv(a->x, a->pf1, a->y1, a->conv_ws, B, C, C, hw, stream);
    if (rc) return rc;
    rc = sc_bn_act_forward(a->y1, nullptr, a->g1, a->b1, a->a1, a->st1, a->st1 + (size_t)G * C, a->rm1, a->rv1, a->nt1, a->bn_ws, B, C, HW, 1,
                           a->training, G, a->eps1, a->mom1, stream);
    if (rc) return rc;
    rc = conv(a->a1, a->pf2, a->y2, a->conv_ws, B, C, C, hw, stream);
    if (rc) return rc;
    return sc_bn_act_forward(a->y2, a->x, a->g2, a->b2, a->out, a->st2, a->st2 + (size_t)G * C, a->rm2, a->rv2, a->nt2, a->bn_ws, B, C, HW, 1,
                             a->training, G, a->eps2, a->mom2, stream);
}

extern "C" int sc_basic_block_backward(const sc_block_args* a, void* stream) {
    if (!a || a->batch <= 0 || a->channels <= 0 || !a->d_out) return (int)hipErrorInvalidValue;
    const int B = a->batch, C = a->channels, hw = a->hw, HW = hw * hw, G = a->groups;
    auto conv = a->split ? sc_conv3x3_forward_split : sc_conv3x3_forward;
    auto wgrad = a->split ? sc_conv3x3_wgrad_split : sc_conv3x3_wgrad;
    int rc = sc_bn_act_backward(a->d_out, a->y2, a->out, a->g2, a->b2, a->st2, a->st2 + (size_t)G * C, a->bn_ws, a->dy2, a->need_dx ? a->dres : nullptr,
                                a->dgb2, a->dgb2 + C, B, C, HW, 1, a->training, G, stream);
    if (rc) return rc;
    rc = conv(a->dy2, a->pb2, a->da1, a->conv_ws, B, C, C, hw, stream);
    if (rc) return rc;
    if (a->gw2) {
        rc = wgrad(a->dy2, a->a1, a->gw2, a->wgrad_ws, B, C, C, hw, stream);
        if (rc) return rc;
    }
    rc = sc_bn_act_backward(a->da1, a->y1, nullptr, a->g1, a->b1, a->st1, a->st1 + (size_t)G * C, a->bn_ws, a->dy1, nullptr, a->dgb1, a->dgb1 + C, B,
                            C, HW, 1, a->training, G, stream);
    if (rc) return rc;
    if (a->need_dx) {       // dx = conv1^T(dy1) + dres: the residual-branch gradient joins in the convolution's store epilogue (round 5)
        rc = sc_conv3x3_forward_add(a->dy1, a->pb1, a->dres, a->dx, a->conv_ws, B, C, C, hw, a->split, stream);
        if (rc) return rc;
    }
    if (a->gw1) rc = wgrad(a->dy1, a->x, a->gw1, a->wgrad_ws, B, C, C, hw, stream);
    return rc;
}
