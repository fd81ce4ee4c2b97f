import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_host_gpu import make_gen
from test_unet2d_gpu import _vunet
from sdeflow_light_amd import sde_scheme as SS, ops, _lib as L
DEV = "cuda"
torch.manual_seed(0)
gen = make_gen("sgm", _vunet(16, "F"))
B, n = 6, 256
x0 = torch.randn(B, n, device=DEV)
for nog in ("",):
    if nog: os.environ["MSGM_NO_GRAPH_SAMPLER"] = "1"
    for fold in ("",):
        if fold: os.environ["MSGM_NO_GN_FOLD"] = "1"
        else: os.environ.pop("MSGM_NO_GN_FOLD", None)
        for method in ("em", "heun", "rk4"):
            for N in (4, 5, 6, 8):
                gs = SS.GraphedStepSampler(gen, B, n, N, method=method)
                st = gen.base_sde.rng.state.clone()
                a = gs.run(x0).clone()
                gen.base_sde.rng.state.copy_(st)
                fn = {"em": SS.euler_maruyama_sampler, "heun": SS.heun_sampler, "rk4": SS.rk4_stratonovich_sampler}[method]
                b = fn(gen, x0, num_steps=N, keep_all_samples=False).to(DEV)
                print("nograph" if nog else "graph", "nofold" if fold else "fold", method, N, float((a - b).norm() / b.norm()), flush=True)
