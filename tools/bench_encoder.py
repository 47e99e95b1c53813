#!/usr/bin/env python3
"""Frozen encoder trunk (EncoderResnet.forward_pyramid, eval mode, no grad): eager MIOpen launches against the HIP-graph replay
(neural-jacobian-field_amd/encoder.py), wall time per call for the shapes the reference uses -- one 256 x 256 image (a frame /
a control step), the seven context images of a training batch, one 480 x 640 notebook frame.  Prints one JSON object."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import __graft_entry__ as entry
    entry.build()
    from neural_jacobian_field_amd.config import model_cfg_from_dict
    from neural_jacobian_field_amd.encoder import EncoderResnet
    from neural_jacobian_field_amd.model import Model
    dev = torch.device("cuda:0")
    enc = Model(model_cfg_from_dict({"action_dim": 8})).encoder.to(dev).eval().requires_grad_(False)
    out = {}
    for name, shape in (("1 x 256 x 256", (1, 3, 256, 256)), ("7 x 256 x 256", (7, 3, 256, 256)), ("1 x 480 x 640", (1, 3, 480, 640))):
        imgs = [torch.rand(shape, device=dev) for _ in range(4)]
        row = {}
        for mode in ("eager", "graph"):
            EncoderResnet._graph_disabled = mode == "eager"
            EncoderResnet._graph_states.pop(enc, None)
            with torch.no_grad():
                for i in range(6):
                    enc.forward_pyramid(imgs[i % 4])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 200
                for i in range(n):
                    enc.forward_pyramid(imgs[i % 4])
                torch.cuda.synchronize()
                row[mode + "_ms"] = round(1e3 * (time.perf_counter() - t0) / n, 4)
        out[name] = row
    EncoderResnet._graph_disabled = False
    print(json.dumps(out))


if __name__ == "__main__":
    main()
