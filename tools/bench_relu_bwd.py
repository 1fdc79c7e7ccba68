#!/usr/bin/env python3
"""STE backward with the folded ReLU gate (qs_quant_ste_relu_bwd), and the matching forward, on the post-residual activation shapes of a ResNet-50
step: fp32 gradient, fp32 ReLU input, fp32 out -- three fp32 streams, 12 B/elem (development tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from qsparse_amd import _hip


def t_us(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3


SHAPES = ((64, 256, 56, 56), (64, 512, 28, 28), (64, 1024, 14, 14), (64, 2048, 7, 7), (256, 256, 56, 56), (256, 1024, 14, 14))
if "--b256" in sys.argv:      # the bf16 sites inside ResNet-50's bottlenecks at batch 256 (and the fp32 ones behind the residual adds)
    SHAPES = ((256, 64, 112, 112), (256, 64, 56, 56), (256, 128, 56, 56), (256, 128, 28, 28), (256, 256, 28, 28), (256, 256, 14, 14),
              (256, 512, 14, 14), (256, 512, 7, 7), (256, 256, 56, 56), (256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7))
for shape in SHAPES:
    for xdt in (torch.float32, torch.bfloat16):
        for cl in (False, True):
            C = shape[1]
            nrot = 3
            xs = [torch.randn(shape, device="cuda").to(xdt) for _ in range(nrot)]
            gs = [torch.randn(shape, device="cuda") for _ in range(nrot)]
            if cl:
                xs = [x.contiguous(memory_format=torch.channels_last) for x in xs]
                gs = [g.contiguous(memory_format=torch.channels_last) for g in gs]
            scale = torch.full((1, 1), 0.1, device="cuda")
            mask = (torch.rand(C, device="cuda") > 0.5)
            turn = [0]

            def run():
                turn[0] += 1
                i = turn[0] % nrot
                _hip.ste_relu_bwd(gs[i], xs[i], scale, False, -8.0, 7.0, mask, 1)

            def fwd():      # the forward of the same site: y = Q(max(x, 0) * mask), fp32 out
                turn[0] += 1
                _hip.quant_fwd("scaler", xs[turn[0] % nrot], scale, -1, torch.float32, chan_mask=mask, mask_channel_index=1,
                               pre_relu=True)

            gates = [_hip.quant_fwd("scaler", x, scale, -1, torch.float32, chan_mask=mask, mask_channel_index=1, pre_relu=True,
                                    want_gate=True)[2] for x in xs]

            def run_gate():     # the same backward from the gate bitmap the forward recorded (1 bit instead of x per element)
                turn[0] += 1
                i = turn[0] % nrot
                _hip.ste_relu_bwd(gs[i], None, scale, False, -8.0, 7.0, mask, 1, gate=gates[i])

            def fwd_gate():
                turn[0] += 1
                _hip.quant_fwd("scaler", xs[turn[0] % nrot], scale, -1, torch.float32, chan_mask=mask, mask_channel_index=1,
                               pre_relu=True, want_gate=True)

            us = t_us(run)
            n, eb = xs[0].numel(), xs[0].element_size()
            nbytes = n * (4 + eb * 2)
            usf = t_us(fwd)
            nbf = n * (4 + eb)
            usg, usfg = t_us(run_gate), t_us(fwd_gate)
            os.environ["QS_RELU_BWD_U"] = "2"            # two groups per lane in the narrowing gate kernels (A/B)
            usg2 = t_us(run_gate)
            os.environ["QS_RELU_BWD_U"] = "1"
            print(f"{str(shape):20s} x {str(xdt)[6:]:8s} {'channels_last' if cl else 'nchw':13s} bwd {us:7.1f} us {nbytes / us / 1e3:6.0f} GB/s"
                  f"   fwd {usf:7.1f} us {nbf / usf / 1e3:6.0f} GB/s   | gate: bwd {usg:7.1f} us {n * (4.125 + eb) / usg / 1e3:6.0f} GB/s"
                  f"   fwd {usfg:7.1f} us {n * (4.125 + eb) / usfg / 1e3:6.0f} GB/s   | U=2 gate bwd {usg2:7.1f} us", flush=True)
