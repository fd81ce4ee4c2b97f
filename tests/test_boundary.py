"""CPU-side checks of the drop-in boundary: the C-ABI library loads and
exports every symbol include/msgm_hip.h declares (no compute calls without a
GPU), the product never imports the oracle, the host mirrors keep the
reference's constructor signatures / state_dict keys / error behaviour, and
the product fails loudly without a GPU."""
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "msgm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(msgm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from sdeflow_light_amd import _lib
    L = _lib.lib()
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in msgm_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert L.msgm_version() >= 100
    assert L.msgm_error_string(-2).decode().startswith("unsupported")
    # pure-host queries are callable without a GPU
    assert L.msgm_mlp_num_params(2, 0) == 33794
    assert L.msgm_mlp_num_params(2, 1) == 33922
    assert L.msgm_mlp_ssm_workspace(2, 0) == 256 * 33796 * 4    # 256 slabs of n_params + 1 (loss) floats, pitch rounded up to 4


def test_product_never_imports_oracle_or_reference():
    pkg = os.path.join(ROOT, "sdeflow_light_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert not re.search(r"""["']/root/reference""", src), f        # may be cited in comments, never opened


def test_kernels_refuse_cpu_tensors():
    from sdeflow_light_amd import ops, _lib
    st = _lib.sde_struct(_lib.SDE_SGM, 0.1, 20.0, 1.0, 1e-3)
    with pytest.raises(_lib.MsgmError, match="no CPU fallback"):
        ops.perturb_vp(torch.zeros(4, 2), st, u=torch.zeros(4), eps=torch.zeros(4, 2))
    with pytest.raises(_lib.MsgmError):
        ops.row_norm(torch.zeros(4, 2))


def test_mlp_state_dict_keys_and_signature():
    from sdeflow_light_amd.NN import MLP
    m = MLP(input_dim=2, index_dim=1, hidden_dim=128, premodule=None)
    assert list(m.state_dict().keys()) == [f"main.{i}.{w}" for i in (0, 2, 4, 6) for w in ("weight", "bias")]
    assert m.state_dict()["main.0.weight"].shape == (128, 3)
    m = MLP(input_dim=6, premodule="NormalizeLogRadius")
    assert m.state_dict()["main.0.weight"].shape == (128, 8) and m.state_dict()["main.6.weight"].shape == (6, 128)
    assert list(inspect.signature(MLP.__init__).parameters)[1:] == ["input_dim", "index_dim", "hidden_dim", "act", "premodule"]
    with pytest.raises(AssertionError):
        MLP(2, premodule="bogus")
    with pytest.raises(Exception, match="no CPU fallback|device"):
        m(torch.zeros(3, 6), torch.zeros(3))              # CPU tensors: loud failure, no eager fallback


def test_reverse_sde_object_graph_and_state_dict():
    from sdeflow_light_amd.NN import MLP
    from sdeflow_light_amd.SDEs import SGMsde, MSGMsde, PluginReverseSDE, forward_SDE
    T = torch.nn.Parameter(torch.FloatTensor([1.0]), requires_grad=False)
    base = SGMsde(beta_min=0.1, beta_max=20.0, t_epsilon=1e-3, T=T, num_steps_forward=16, device="cpu")
    gen = PluginReverseSDE(base, MLP(2), T, vtype="rademacher", debias=False, ssm_intT=False, deviceReverseSDE="cpu")
    keys = list(gen.state_dict().keys())
    assert "T" in keys and "base_sde.T" in keys and "a.main.0.weight" in keys
    assert base.name_SDE == "SGM" and base.sparseTensor is False and base.IJK() == (None, None, None)
    assert float(base.beta(torch.tensor(0.5))) == pytest.approx(10.05)
    ms = MSGMsde(torch.randn(32, 6), T=T, denseTensor=False, norm_map="log", num_steps_forward=4)
    I, J, K = ms.IJK()
    assert I.tolist()[:4] == [0, 1, 1, 2] and J.tolist()[:4] == [1, 0, 2, 1] and K.tolist()[:4] == [0, 0, 1, 1]   # App. B #11
    assert ms.name_SDE == "MSGM_sparseTenslogNorm" and ms.norm_correction and ms.sparseTensor
    md = MSGMsde(torch.randn(32, 4), T=T, denseTensor=True)
    assert float(torch.trace(md.L_G)) == pytest.approx(-2.0, abs=1e-4)              # App. B #10
    assert isinstance(forward_SDE(ms, T).base_sde, MSGMsde)
    with pytest.raises(Exception):
        PluginReverseSDE(base, MLP(2), T, ssm_intT=True)


def test_integrator_signatures_match_reference():
    from sdeflow_light_amd import sde_scheme as S
    want = ["sde", "x_0", "num_steps", "lmbd", "keep_all_samples", "samplesToKeep", "include_t0", "T_", "norm_correction"]
    for fn in (S.euler_maruyama_sampler, S.heun_sampler, S.rk4_stratonovich_sampler):
        sig = inspect.signature(fn)
        assert list(sig.parameters)[:9] == want
        assert sig.parameters["num_steps"].default == 1000 and sig.parameters["keep_all_samples"].default is True
        assert sig.parameters["T_"].default == -1 and sig.parameters["include_t0"].default is False


def test_shard_rows_cover_everything():
    from sdeflow_light_amd.parallel import shard_rows
    for n in (1, 7, 8192, 65536 + 3):
        for w in (1, 2, 3, 8):
            cuts = [shard_rows(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(e - b for b, e in cuts) - min(e - b for b, e in cuts) <= 1
