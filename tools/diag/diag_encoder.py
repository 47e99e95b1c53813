import os, sys, torch
ROOT="/root/repo" if os.path.exists("/root/repo/oracle") else os.getcwd()
sys.path[:0]=[ROOT, os.path.join(ROOT,"oracle")]
import __graft_entry__ as g_; g_.build()
import njf_oracle as orc, parity_harness as ph
from neural_jacobian_field_amd import synthetic
from neural_jacobian_field_amd.config import model_cfg_from_dict
from neural_jacobian_field_amd.model import Model
rel=lambda a,b:((a.double().cpu()-b.double().cpu()).abs().max()/(b.double().cpu().abs().max()+1e-30)).item()
dev=torch.device("cuda:0")
for seed,(H,W) in ((4,(16,16)),(0,(16,16)),(4,(64,64))):
    full=synthetic.seeded_state_dict(synthetic.model_shapes("jacobian_mlp",8),seed=seed)
    model=Model(model_cfg_from_dict({"action_dim":8,"rendering":{"num_proposal_samples":[32],"num_nerf_samples":32},"action_decoder":{"name":"jacobian_mlp"}}))
    model.load_state_dict(full,strict=True); model.to(dev).eval()
    image=torch.rand(2,3,H,W,generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        fh=model.encoder.forward(image.to(dev))
        fo=orc.encoder_features({k[len("encoder."):]:v for k,v in full.items() if k.startswith("encoder.")},image)
        fo64=orc.encoder_features({k[len("encoder."):]:(v.double() if v.is_floating_point() else v) for k,v in full.items() if k.startswith("encoder.")},image.double())
        lat=model.encoder._latents(image.to(dev))
    print(f"seed {seed} {H}x{W}: HIP(MIOpen) vs oracle fp32 {rel(fh,fo):.2e}; oracle fp32 vs fp64 {rel(fo,fo64):.2e}; HIP vs fp64 {rel(fh,fo64):.2e}; max|f| {fo.abs().max():.3g}")
    c0=0
    for i,l in enumerate(lat):
        c=l.shape[1]; print(f"   level {i} channels {c0}:{c0+c}: HIP vs oracle {rel(fh[:,c0:c0+c],fo[:,c0:c0+c]):.2e}"); c0+=c
