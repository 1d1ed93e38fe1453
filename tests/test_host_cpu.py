"""CPU-side checks of the host package: C ABI export table, state_dict contract, engine plan, host schedule math.
No compute kernels are launched here (there is no GPU in this tier)."""
import ctypes
import math
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    """(product symbols, probe-only symbols) declared by include/vdiff_hip.h; the latter sit inside #ifdef VD_PROBES"""
    hdr = open(os.path.join(ROOT, "include", "vdiff_hip.h")).read()
    probe_txt = "".join(re.findall(r"#ifdef VD_PROBES(.*?)#endif", hdr, re.S))
    prod_txt = re.sub(r"#ifdef VD_PROBES.*?#endif", "", hdr, flags=re.S)
    sym = lambda txt: sorted(set(re.findall(r"\b(vd_[a-z0-9_]+)\s*\(", txt)))
    return sym(prod_txt), sym(probe_txt)


def test_c_abi_exports_every_declared_symbol():
    from v_diffusion import _hip
    declared, probe_only = _header_symbols()
    assert declared == sorted(_hip.EXPORTS), "binding table and header disagree"
    assert probe_only == sorted(_hip.PROBE_EXPORTS)
    if not os.path.exists(_hip.LIB_PATH):
        pytest.fail(f"{_hip.LIB_PATH} missing: run __graft_entry__.build()")
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"
    lib.vd_version.restype = ctypes.c_int
    assert lib.vd_version() == 100


def test_product_library_has_no_probe_or_experiment_code():
    """Round-2 review: the shipped .so honoured VD_WINO_EXP (timing variants that compute WRONG results) and exported
    vd_wino_set_probe.  Those live in libvdiff_hip_probe.so now (-DVD_PROBES, tests/probe/ only): the product library must not
    export the probe entry point, must not read the probe / experiment knobs, and must not contain a PROBE / EXP instantiation."""
    from v_diffusion import _hip
    assert os.path.basename(_hip.LIB_PATH) == "libvdiff_hip.so" or os.environ.get("VDIFF_HIP_LIB")
    prod_path = os.path.join(ROOT, "v-diffusion-torch_amd", "lib", "libvdiff_hip.so")
    probe_path = os.path.join(ROOT, "v-diffusion-torch_amd", "lib", "libvdiff_hip_probe.so")
    prod = ctypes.CDLL(prod_path)
    _, probe_only = _header_symbols()
    for name in probe_only:
        assert not hasattr(prod, name), f"product library exports {name}"
    blob = open(prod_path, "rb").read()
    for knob in (b"VD_WINO_EXP", b"VD_GEMM_PROBE", b"VD_WINO_PROBE_LIGHT", b"VD_WGRAD_EXP"):
        assert knob not in blob, f"product library reads {knob.decode()}"
    # mangled template arguments <TW, NS, STATS, PROBE, EXP> of wino_conv_kernel: ...Lb<stats>ELb<probe>ELi<exp>E
    inst = set(re.findall(rb"wino_conv_kernelILi\d+ELi\d+ELb[01]ELb([01])ELi(\d+)E", blob))
    assert inst == {(b"0", b"0")}, f"probe / experiment instantiations in the product library: {sorted(inst)}"
    if os.path.exists(probe_path):                        # the probe build keeps them (and nothing in the package loads it)
        pblob = open(probe_path, "rb").read()
        assert b"VD_WINO_EXP" in pblob and hasattr(ctypes.CDLL(probe_path), "vd_wino_set_probe")
    pkg = os.path.join(ROOT, "v-diffusion-torch_amd", "v_diffusion")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            code = [l for l in open(os.path.join(pkg, fn)) if not l.lstrip().startswith("#")]
            assert not any("libvdiff_hip_probe" in l for l in code), fn       # (comments may name it)
    assert "libvdiff_hip_probe" not in open(os.path.join(ROOT, "bench.py")).read()


def test_gemm_desc_layout_matches_header(tmp_path):
    """the ctypes mirror of vd_gemm_desc has the layout gcc gives the C struct"""
    import subprocess
    from v_diffusion._hip import GemmDesc
    fields = [f[0] for f in GemmDesc._fields_]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "vdiff_hip.h"\nint main(){printf("%zu", sizeof(vd_gemm_desc));' + \
        "".join(f'printf(" %zu", offsetof(vd_gemm_desc, {f}));' for f in fields) + "return 0;}"
    src = tmp_path / "layout.c"
    src.write_text(prog)
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    nums = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert nums[0] == ctypes.sizeof(GemmDesc)
    assert nums[1:] == [getattr(GemmDesc, f).offset for f in fields]


@pytest.mark.parametrize("cfgname", ["CIFAR_COND", "CIFAR_UNCOND", "CELEBA"])
def test_state_dict_contract_cpu(golden_dir, cfgname):
    from oracle import cases
    from oracle.unet_ref import param_shapes
    import v_diffusion
    cfg = getattr(cases, cfgname)
    m = v_diffusion.UNet(**cfg)
    shapes = param_shapes(cfg)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(s)) for k, s in shapes.items()]
    if cfgname != "CIFAR_UNCOND":
        g = np.load(os.path.join(golden_dir, "unet_cifar10_cond.npz" if cfgname == "CIFAR_COND" else "unet_celeba.npz"))
        assert [str(n) for n in g["grad_names"]] == [k for k, _ in m.named_parameters()]      # reference parameters() order


def test_use_xformers_flag_falls_back_to_the_plain_attention_block(capsys):
    """SURVEY 8a row 17 (reference unet.py:192-199): without the xformers package the reference prints a message, resets the flag and
    builds the plain AttentionBlock -- conv-shaped `proj_in` / `proj_out` parameters, none of XFormersAttentionBlock's
    `to_q` / `to_k` / `to_v` linears (unet.py:84-103).  Same parameter set, same state_dict as use_xformers=False."""
    from oracle import cases
    import v_diffusion
    cfg = cases.TINY["tinyA"]["cfg"]
    ref = v_diffusion.UNet(**cfg)
    capsys.readouterr()
    m = v_diffusion.UNet(**cfg, use_xformers=True)
    assert "xFormers not available! Resetting to False." in capsys.readouterr().out
    keys = list(m.state_dict().keys())
    assert keys == list(ref.state_dict().keys())
    assert [tuple(v.shape) for v in m.state_dict().values()] == [tuple(v.shape) for v in ref.state_dict().values()]
    assert any(k.endswith("proj_in.weight") for k in keys) and any(k.endswith("proj_out.weight") for k in keys)
    assert not any(("to_q" in k) or ("to_k" in k) or ("to_v" in k) for k in keys)
    pin = next(v for k, v in m.state_dict().items() if k.endswith("proj_in.weight"))
    assert pin.ndim == 4 and pin.shape[2:] == (1, 1)                       # a 1x1 convolution, not a Linear


def test_engine_plan_concat_wiring():
    from oracle import cases
    import v_diffusion
    from v_diffusion.engine import UNetEngine
    for cfg in (cases.CIFAR_COND, cases.CELEBA, cases.TINY["tinyB"]["cfg"]):
        m = v_diffusion.UNet(**cfg)
        e = UNetEngine(m)
        nrb, L = cfg["num_res_blocks"], len(cfg["ch_multipliers"])
        assert len(e.pushes) == 1 + L * nrb + (L - 1)                       # reference: every down output is pushed
        assert sum(b.consumes for b in e.plan) == len(e.pushes)             # and popped exactly once
        for k, (_, cs, consumer, ch) in enumerate(e.pushes):
            assert e.plan[consumer].cin == ch + cs
        # pops are LIFO
        order = [b.src_hs for b in e.plan if b.consumes]
        assert order == sorted(order, reverse=True)


def test_lecun_init_statistics():
    from v_diffusion.modules import Conv2d, Linear
    torch.manual_seed(0)
    c = Conv2d(64, 128, 3, 1, 1)
    std = c.weight.std().item()
    # truncated normal at +-2 sigma has std 0.8796; reference applies no variance correction (modules.py:25-35)
    assert abs(std - 0.8796 / (64 * 9) ** 0.5) < 0.01 / (64 * 9) ** 0.5 * 3
    assert float(c.bias.abs().max()) == 0 and float(c.weight.abs().max()) <= 2.0 / (64 * 9) ** 0.5 + 1e-6
    assert float(Linear(8, 8, init_scale=0.).weight.abs().max()) == 0


def test_host_schedule_and_posteriors_vs_golden(golden_dir):
    import v_diffusion
    from v_diffusion.diffusion import logsnr_to_posterior, logsnr_to_posterior_ddim
    g = np.load(os.path.join(golden_dir, "tables.npz"))
    for sched in ("cosine", "linear", "sigmoid", "legacy"):
        f = v_diffusion.get_logsnr_schedule(sched, -20.0, 20.0)
        for T in (8, 50, 250):
            grid = torch.arange(T + 1, dtype=torch.float64) / T
            if sched == "linear":
                grid = grid.clamp(1e-6, 1 - 1e-6)
            np.testing.assert_allclose(f(grid).numpy(), g[f"logsnr_{sched}_{T}"], rtol=1e-12, atol=1e-9)
    for T in (8, 50, 250):
        l = torch.from_numpy(g[f"logsnr_cosine_{T}"])
        ls, lt = l[:-1].float(), l[1:].float()
        c1, c2, lv = logsnr_to_posterior_ddim(ls, lt, 0.0)
        np.testing.assert_allclose(np.stack([c1.numpy(), c2.numpy()]), g[f"ddim_{T}"], rtol=1e-6)
        assert float(lv) == -np.inf
        for vt, frac in (("fixed_large", None), ("fixed_small", None), ("fixed_medium", 0.3)):
            c1, c2, lv = logsnr_to_posterior(ls, lt, vt, frac)
            np.testing.assert_allclose(np.stack([c1.numpy(), c2.numpy(), lv.numpy()]), g[f"ddpm_{vt}_{T}"], rtol=1e-6)
    # DDIM(eta=1) == DDPM fixed_small (reference self-check diffusion.py:594-614)
    a = logsnr_to_posterior_ddim(ls, lt, 1.0)
    b = logsnr_to_posterior(ls, lt, "fixed_small")
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_step_coefficients_match_reference_tables(golden_dir):
    import v_diffusion
    g = np.load(os.path.join(golden_dir, "tables.npz"))
    T = 50
    gd = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine"), T, "v", "fixed_medium", "snr_trunc", "mse",
                                       intp_frac=0.3, w_guide=1.0)
    for step in (0, 1, 25, 49):
        k, _ = gd._step_coefs(step, use_ddim=True)
        np.testing.assert_allclose(k[3:5], g["ddim_50"][:, step], rtol=1e-6)
        assert k[5] == 0.0 and k[6] == 1.0
        k, _ = gd._step_coefs(step, use_ddim=False)
        np.testing.assert_allclose(k[3:5], g["ddpm_fixed_medium_50"][:2, step], rtol=1e-6)
        if step > 0:
            np.testing.assert_allclose(k[5], np.exp(0.5 * g["ddpm_fixed_medium_50"][2, step]), rtol=1e-5)
        lt = np.float32(g["logsnr_cosine_50"][step + 1])
        np.testing.assert_allclose(k[0], np.sqrt(1 / (1 + np.exp(-np.float64(lt)))), rtol=1e-6)


def test_per_sample_posterior_api_matches_oracle():
    """p_mean_var / q_posterior_mean_var(_ddim): the reference's tensor-valued API (device-agnostic tensor expressions)"""
    import v_diffusion
    from oracle import diffusion_ref as dref, detrand
    B = 4
    xt, out = detrand.normal("pm_x", (B, 3, 8, 8), 1), detrand.normal("pm_o", (B, 6, 8, 8), 2)
    sched = v_diffusion.get_logsnr_schedule("cosine")
    s = torch.tensor([0.0, 0.1, 0.5, 0.9], dtype=torch.float64)
    t = s + 0.1
    ls, lt = sched(s).float().reshape(-1, 1, 1, 1), sched(t).float().reshape(-1, 1, 1, 1)
    for mot in ("v", "x0", "eps", "both"):
        o = out if mot == "both" else out[:, :3]
        for ddim in (True, False):
            gd = v_diffusion.GaussianDiffusion(sched, 8, mot, "fixed_medium", "snr_trunc", "mse", intp_frac=0.3)
            mean, lv, pred = gd.p_mean_var(o, xt, ls, lt, clip_denoised=True, return_pred=True, use_ddim=ddim)
            px0 = dref.predictions(mot, xt, o, lt)[0].clamp(-1, 1)
            c1, c2, rv = dref.ddim_coefs(ls, lt) if ddim else dref.ddpm_coefs(ls, lt, "fixed_medium", 0.3)
            torch.testing.assert_close(pred, px0, rtol=1e-6, atol=1e-6)
            torch.testing.assert_close(mean, c1 * xt + c2 * px0, rtol=1e-6, atol=1e-6)
            if not ddim:
                torch.testing.assert_close(lv, rv, rtol=1e-6, atol=1e-6)


def test_cpu_tensors_fail_loudly():
    import v_diffusion
    from oracle.cases import TINY, make_inputs
    case = TINY["tinyC"]
    m = v_diffusion.UNet(**case["cfg"])
    x, t, y = make_inputs(case["cfg"], 2, case["R"], case["label"])
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(x, t, y)
    gd = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine"), 8, "v", "fixed_large", "snr_trunc", "mse")
    with pytest.raises(RuntimeError, match="no CPU path"):
        gd.train_loss(m, x, t, y)
    with pytest.raises(RuntimeError, match="no CPU path"):
        gd.p_sample(m, (2, 3, 8, 8), device="cpu")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "v-diffusion-torch_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"


def test_host_posteriors_x0eps_and_eta_vs_golden(golden_dir):
    """logsnr_to_posterior(_ddim) with x0eps_coef=True and 0 < eta < 1 against the reference's numbers"""
    import v_diffusion
    from v_diffusion.diffusion import logsnr_to_posterior, logsnr_to_posterior_ddim
    g = np.load(os.path.join(golden_dir, "ext_tables.npz"))
    f = v_diffusion.get_logsnr_schedule("cosine")
    for T in (8, 50):
        l = f(torch.arange(T + 1, dtype=torch.float64) / T)
        ls, lt = l[:-1].float(), l[1:].float()
        for vt, frac in (("fixed_large", None), ("fixed_small", None), ("fixed_medium", 0.3)):
            c = logsnr_to_posterior(ls, lt, vt, frac, x0eps_coef=True)
            np.testing.assert_allclose(np.stack([v.numpy() for v in c]), g[f"ddpm_x0eps_{vt}_{T}"], rtol=1e-6)
        c = logsnr_to_posterior_ddim(ls, lt, 0.0, x0eps_coef=True)
        np.testing.assert_allclose(np.stack([c[0].numpy(), c[1].numpy()]), g[f"ddim_x0eps_{T}"], rtol=1e-6)
        for eta in (0.5, 0.2):
            for xe in (False, True):
                c = logsnr_to_posterior_ddim(ls, lt, eta, x0eps_coef=xe)
                np.testing.assert_allclose(np.stack([v.numpy() for v in c]), g[f"ddim_eta{eta}_{'x0eps' if xe else 'xt'}_{T}"], rtol=1e-6)


def test_reference_self_checks():
    """The reference's own in-file checks (diffusion.py:583-676, run there as ``python diffusion.py``), on this build's
    host functions: the eps/x0 and x_t/x0 forms of the posteriors agree, eta = 1 DDIM is the fixed_small posterior, the
    "legacy" schedule reproduces the DDPM linear-beta alphas, and the cosine schedule inverts/round-trips."""
    import v_diffusion
    from v_diffusion.diffusion import logsnr_to_posterior, logsnr_to_posterior_ddim
    sched = v_diffusion.get_logsnr_schedule("cosine")
    logsnr = sched(torch.linspace(0, 1, 1001))
    ls, lt = logsnr[:-1], logsnr[1:]
    # test_logsnr_to_posterior (:583-591)
    c1, c2, _ = logsnr_to_posterior(ls, lt, "fixed_small")
    e1, e2, _ = logsnr_to_posterior(ls, lt, "fixed_small", x0eps_coef=True)
    logr = lt - ls
    assert torch.allclose(c1 * torch.sigmoid(-lt).sqrt(), e1)
    assert torch.allclose(c2 + torch.sigmoid(ls).sqrt() * logr.exp(), e2)
    # test_logsnr_to_posterior_ddim (:594-614)
    for a, b in zip(logsnr_to_posterior(ls, lt, "fixed_small"), logsnr_to_posterior_ddim(ls, lt, eta=1.)):
        assert torch.allclose(a, b)
    c1, c2, _ = logsnr_to_posterior_ddim(ls, lt, eta=0.5)
    e1, e2, _ = logsnr_to_posterior_ddim(ls, lt, eta=0.5, x0eps_coef=True)
    assert torch.allclose(c1 * torch.sigmoid(-lt).sqrt(), e1)
    # (the reference prints the second identity with alpha_s, :613, which is False there too; the consistent form uses alpha_t)
    assert not torch.allclose(c2 + torch.sigmoid(ls).sqrt() * c1, e2)
    assert torch.allclose(c2 + torch.sigmoid(lt).sqrt() * c1, e2)
    # test_legacy (:617-624): continuous version of the linear-beta schedule
    t = torch.linspace(0, 1, 1000, dtype=torch.float32)
    alphas = torch.sigmoid(v_diffusion.get_logsnr_schedule("legacy")(t))
    ref = torch.cumprod(1 - torch.linspace(0.0001, 0.02, 1000), dim=0)
    # the reference prints 1.976e-3 / 3.794e-3 for these two errors (run here against /root/reference)
    assert abs(float((alphas - ref).abs().max()) - 1.9765e-3) < 2e-6 and abs(float(((alphas - ref) / ref).abs().max()) - 3.7940e-3) < 2e-6
    # test_schedule (:627-676): rescale=True rewrites t <- logsnr2t(logsnr) in place; cosine/sine/cotangent identities
    f = v_diffusion.get_logsnr_schedule("cosine", rescale=True)
    idx = np.linspace(0, 1000, 50).astype(np.int64)
    t = torch.linspace(0, 1, 1001, dtype=torch.float32)[idx]
    t_in = t.clone()
    l = f(t)
    at = torch.atan(l.clamp(-20, 20).mul(-0.5).exp()).div(0.5 * math.pi)
    assert torch.allclose(t, at, atol=1e-6)
    assert not torch.equal(t, t_in)                    # endpoints moved from [0, 1] to [t(logsnr_max), t(logsnr_min)]
    li = l[1:-1]
    assert torch.allclose(torch.sigmoid(li).sqrt(), li.exp().sqrt() * torch.sigmoid(-li).sqrt())
    assert torch.allclose(torch.sigmoid(-li).sqrt(), li.neg().exp().sqrt() * torch.sigmoid(li).sqrt())
    assert torch.allclose(li.mul(0.5).exp(), torch.sigmoid(li).sqrt() * torch.sigmoid(-li).rsqrt(), rtol=1e-4)


def test_step_coefficients_fold_the_eps_form(golden_dir):
    """x0eps_coef=True: the sampler's (x_t, x0_hat) weights reproduce c1*eps + c2*x0_hat with eps re-derived from x0_hat"""
    import v_diffusion
    g = np.load(os.path.join(golden_dir, "ext_tables.npz"))
    T = 8
    sched = v_diffusion.get_logsnr_schedule("cosine")
    gd = v_diffusion.GaussianDiffusion(sched, T, "v", "fixed_medium", "snr_trunc", "mse", intp_frac=0.3, x0eps_coef=True)
    l = sched(torch.arange(T + 1, dtype=torch.float64) / T).float().double()
    for step in range(T):
        k, _ = gd._step_coefs(step, use_ddim=False)
        c1, c2 = g["ddpm_x0eps_fixed_medium_8"][0, step].astype(np.float64), g["ddpm_x0eps_fixed_medium_8"][1, step].astype(np.float64)
        alpha, sigma = float(torch.sigmoid(l[step + 1]).sqrt()), float(torch.sigmoid(-l[step + 1]).sqrt())
        xt, x0h = 0.37, -0.81
        want = c1 * (xt - alpha * x0h) / sigma + c2 * x0h
        assert abs(k[3] * xt + k[4] * x0h - want) <= 1e-5 * max(1.0, abs(want))


def test_public_diffusion_helpers_vs_golden(golden_dir):
    """module-level helpers of reference diffusion.py:19-39,206-250 and GaussianDiffusion.from_model_out_to_pred (:466-490)
    against the reference's numbers (oracle/make_goldens_r2.py).  Tensor-shape / elementwise glue: device-agnostic."""
    import v_diffusion
    from v_diffusion import diffusion as D
    from oracle.make_goldens_r2 import helper_inputs
    g = np.load(os.path.join(golden_dir, "r2_helpers.npz"))
    x0, eps, out, out2, l = helper_inputs()
    xt = D.q_sample(x0, l, eps=eps)
    tol = dict(rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(xt.numpy(), g["q_sample"], **tol)
    xt = torch.from_numpy(g["q_sample"])
    m, lv = D.q_mean_var(x0, l)
    np.testing.assert_allclose(m.numpy(), g["q_mean"], **tol)
    np.testing.assert_allclose(lv.numpy(), g["q_logvar"], **tol)
    big = dict(rtol=3e-6, atol=3e-6 * 500)              # 1/alpha reaches e^6 at logsnr = -12
    np.testing.assert_allclose(D.pred_x0_from_eps(xt, out, l).numpy(), g["pred_x0_from_eps"], **big)
    np.testing.assert_allclose(D.pred_x0_from_x0eps(xt, out2, l).numpy(), g["pred_x0_from_x0eps"], **big)
    np.testing.assert_allclose(D.pred_eps_from_x0(xt, out, l).numpy(), g["pred_eps_from_x0"], rtol=3e-6, atol=3e-3)
    np.testing.assert_allclose(D.pred_v_from_x0eps(x0, eps, l).numpy(), g["pred_v_from_x0eps"], **tol)
    np.testing.assert_allclose(D.pred_v_from_x0(xt, out, l).numpy(), g["pred_v_from_x0"], rtol=3e-6, atol=3e-3)
    np.testing.assert_allclose(D.pred_x0_from_v(xt, out, l).numpy(), g["pred_x0_from_v"], **tol)
    np.testing.assert_allclose(D.pred_eps_from_v(xt, out, l).numpy(), g["pred_eps_from_v"], **tol)
    for mot in ("v", "x0", "eps", "both"):
        gd = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine"), 8, mot, "fixed_large", "snr_trunc", "mse")
        d = gd.from_model_out_to_pred(xt, out2 if mot == "both" else out, l)
        for key in ("constant", "snr", "snr_1plus"):
            np.testing.assert_allclose(d[key].numpy(), g[f"fmo_{mot}_{key}"], rtol=5e-6, atol=5e-3 if mot != "v" else 5e-6, err_msg=f"{mot}/{key}")
        assert torch.equal(d["snr_trunc"][0], d["constant"]) and torch.equal(d["snr_trunc"][1], d["snr"])
    r = D.repeat_along_dim(x0[:, :, 0, 0], 3, dim=0)
    assert np.array_equal(r.numpy(), g["repeat_dim0"]) and r.is_contiguous()
    assert np.array_equal(D.repeat_along_dim(x0[:, :, 0, 0], 2, dim=1).numpy(), g["repeat_dim1"])
    sl = D.slice_along_batch(r, 3)
    assert len(sl) == 3 and np.array_equal(sl[0].numpy(), g["slice_0"]) and np.array_equal(sl[2].numpy(), g["slice_2"])
    b = D.broadcast_to([1.0, 2.0, 3.0], x0)
    assert b.shape == (3, 1, 1, 1) and b.dtype == x0.dtype and np.array_equal(b.numpy(), g["broadcast"])


def test_rescaling_schedule_rewrites_t_and_samplers_use_it():
    """get_logsnr_schedule(rescale=...) rewrites t in place (reference :105-109); the sampler hands THAT t to the network
    (:363-374) -- the host side of it: _step_coefs returns the rewritten time."""
    import v_diffusion
    for rescale, expect in ((0.5, lambda t: 0.5 * t), (True, None)):
        f = v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0, rescale=rescale)
        gd = v_diffusion.GaussianDiffusion(f, 8, "v", "fixed_large", "snr_trunc", "mse", w_guide=1.0)
        plain = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), 8, "v", "fixed_large",
                                              "snr_trunc", "mse", w_guide=1.0)
        for step in (0, 3, 7):
            k, tn = gd._step_coefs(step, True)
            k0, t0 = plain._step_coefs(step, True)
            assert k == k0 and t0 == (step + 1) / 8
            if expect is not None:
                assert tn == expect((step + 1) / 8)
            else:                                          # bool: t <- logsnr2t(logsnr(t)) = lerp(t_from, t_to, t)
                l = float(f(torch.tensor([(step + 1) / 8], dtype=torch.float64))[0])
                assert abs(tn - float(np.arctan(np.exp(-0.5 * l)) / (0.5 * np.pi))) < 1e-12 and (step == 3 or tn != (step + 1) / 8)   # (t = 1/2 is the fixed point of the symmetric schedule)


def test_reference_control_plane_names_are_delegated(tmp_path, monkeypatch):
    """v_diffusion.<control-plane name> resolves from a reference checkout named by VDIFF_REFERENCE_ROOT (loaded under the
    alias v_diffusion_ref so its relative imports stay inside it) and fails with a clear ImportError without one."""
    import importlib
    import sys
    import v_diffusion
    monkeypatch.delenv("VDIFF_REFERENCE_ROOT", raising=False)
    monkeypatch.setattr(v_diffusion, "_ref_pkg", None)
    with pytest.raises(ImportError, match="VDIFF_REFERENCE_ROOT"):
        v_diffusion.Trainer
    with pytest.raises(AttributeError):
        v_diffusion.no_such_name
    pkg = tmp_path / "v_diffusion"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("from .train_utils import Trainer\nfrom .utils import seed_all\nDATA_INFO = {'k': 1}\n")
    (pkg / "train_utils.py").write_text("from .utils import seed_all\nclass Trainer:\n    helper = staticmethod(seed_all)\n")
    (pkg / "utils.py").write_text("def seed_all(s):\n    return ('seeded', s)\n")
    monkeypatch.setenv("VDIFF_REFERENCE_ROOT", str(tmp_path))
    try:
        assert v_diffusion.Trainer.helper(3) == ("seeded", 3) and v_diffusion.DATA_INFO == {"k": 1}
        assert v_diffusion.Trainer.__module__ == "v_diffusion_ref.train_utils"
        assert v_diffusion.UNet.__module__ == "v_diffusion.models.unet"          # the hot-path names stay this package's
    finally:
        for k in [k for k in sys.modules if k.startswith("v_diffusion_ref")]:
            del sys.modules[k]
        v_diffusion._ref_pkg = None


def test_no_kernel_spills_to_scratch(tmp_path):
    """hipcc -S of every HIP source for gfx950: no kernel instantiation may spill VGPRs or use scratch memory (round 1
    shipped seven KT = 16 GEMM instantiations with 11-60 spilled VGPRs in their epilogue)."""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from spill_report import kernels, scratch_in_inner_loops
    csrc = os.path.join(ROOT, "v-diffusion-torch_amd", "csrc")
    srcs = [f for f in sorted(os.listdir(csrc)) if f.endswith(".hip")]

    def one(f):
        out = str(tmp_path / (f + ".s"))
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "--cuda-device-only",
                            "-S", os.path.join(csrc, f), "-o", out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return f, kernels(open(out).read())
    with ThreadPoolExecutor(max_workers=4) as pool:
        res = list(pool.map(one, srcs))
    total = 0
    for f, ks in res:
        total += len(ks)
        # (SGPR spills go to VGPR lanes, not to memory.)  No exemptions: round 2 held the persistent Winograd convolution to a
        # weaker rule (2-5 spilled VGPRs outside its K loop); its per-item lane constants are now re-derived per item and its
        # epilogue offsets are 32-bit, so it is spill-free like everything else.
        bad = [k for k in ks if k["spill"] or k["scratch"]]
        assert not bad, f"{f}: kernels with spills / scratch: {bad[:4]}"
    assert total > 90


def test_device_running_statistics_matches_reference_contract():
    """DeviceRunningStatistics == reference RunningStatistics (train_utils.py:30-59) restated here: counts, sums of n * value,
    extract() = sums / count, reset(), keys added on the fly; tensors and plain numbers are both accepted."""
    import torch
    from v_diffusion.trainer import DeviceRunningStatistics

    class Ref:                                                     # the reference semantics, restated
        def __init__(self, **kw):
            self.count, self.stats = 0, {k: (v or 0) for k, v in kw.items()}
        def reset(self):
            self.count = 0
            for k in self.stats:
                self.stats[k] = 0
        def update(self, n, **kw):
            self.count += n
            for k, v in kw.items():
                self.stats[k] = self.stats.get(k, 0) + v
        def extract(self):
            return {k: v / self.count for k, v in self.stats.items()}

    rng = np.random.default_rng(3)
    ref, dev = Ref(loss=None), DeviceRunningStatistics(loss=None)
    for step in range(50):
        B = int(rng.integers(1, 129))
        loss = float(rng.random())
        ref.update(B, loss=loss * B)
        dev.update(B, loss=torch.tensor(loss, dtype=torch.float32).double() * B if step % 2 else loss * B)
        if step == 20:
            ref.update(B, aux=1.5)
            dev.update(B, aux=1.5)
    assert dev.count == ref.count
    got, want = dev.extract(), ref.extract()
    assert got.keys() == want.keys()
    for k in want:
        assert abs(got[k] - want[k]) <= 1e-7 * abs(want[k]), (k, got[k], want[k])
    assert "Count(s): %d" % ref.count in repr(dev)
    ref.reset(); dev.reset()
    assert dev.count == 0 and all(v == 0 for v in dev.stats.values())
    dev.update(2, loss=3.0)
    assert dev.extract()["loss"] == 1.5


# ------------------------------------------------------------------------------------------------------------------------------------
# chain of autograd nodes (models/unet.py::_SegFn): the mechanics on CPU with a stand-in for the engine's backward generator
import sys as _sys
_sys.path.insert(0, os.path.join(ROOT, "tests"))
from chain_stub import ChainStub as _ChainStub      # noqa: E402  (tests/chain_stub.py: shared with the two-rank DDP test)


def test_autograd_chain_hands_gradients_over_segment_by_segment():
    m = _ChainStub()
    for k, p in m.named_parameters():
        p.register_hook(lambda g, k=k: m.log.append(f"grad {k}"))
    x = torch.ones(2, 3, requires_grad=True)
    out = m(x)
    assert torch.equal(out, x.detach() * 2)
    out.sum().backward()
    # every segment's gradients reach their hooks (= DDP's reducer) BEFORE the next segment's kernels are launched
    assert m.log == ["kernels c", "grad c", "kernels b-part", "kernels b", "grad b", "kernels a", "closed", "grad a"], m.log
    assert torch.equal(m.c.grad, torch.full((4,), 6.0)) and torch.equal(m.b.grad, torch.full((2,), 12.0))
    assert torch.equal(m.a.grad, torch.full((3,), 18.0)) and torch.equal(x.grad, torch.full((2, 3), 5.0))
    with pytest.raises(RuntimeError, match="called twice"):
        m.log.clear()
        out2 = m(x)
        out2.sum().backward(retain_graph=True)
        out2.sum().backward()


def test_autograd_chain_two_forwards_one_backward_and_frozen_prefix():
    m = _ChainStub()
    x = torch.ones(2, 3)
    (m(x).sum() + 2 * m(x).sum()).backward()                            # two passes of one engine inside one autograd run
    assert torch.equal(m.c.grad, torch.full((4,), 6.0 + 12.0)) and torch.equal(m.a.grad, torch.full((3,), 18.0 + 36.0))
    assert m.active is None
    # input without gradient and the first segment's parameters frozen: nothing upstream of segment b takes part, the pass ends there
    m2 = _ChainStub()
    m2.a.requires_grad_(False)
    m2(x).sum().backward()
    assert m2.a.grad is None and torch.equal(m2.b.grad, torch.full((2,), 12.0))
    assert "kernels a" not in m2.log and m2.log[-1] == "closed" and m2.active is None


def test_autograd_chain_abandoned_backward_is_closed_not_drained():
    """round-5 advice: a backward that stops between nodes (a parameter hook raised) leaves a suspended pass behind.  Nothing but the autograd
    graph may keep it alive: once the graph is gone the pass closes its generator (the engine's side stream / arena state is released, the
    tape freed), and the next backward starts clean instead of first running the stale pass to completion"""
    import gc
    m = _ChainStub()
    h = m.b.register_hook(lambda g: (_ for _ in ()).throw(RuntimeError("hook failed")))
    x = torch.ones(2, 3)

    def failing_step():                                  # (its own frame: the exception and its traceback -- which reference the pass -- die with it)
        out = m(x)
        try:
            out.sum().backward()
        except RuntimeError as e:
            assert "hook failed" in str(e)
            return True
        return False
    assert failing_step()
    assert m.log == ["kernels c", "kernels b-part", "kernels b"], m.log           # suspended behind segment b; segment a never ran
    h.remove()
    gc.collect()
    # (PyTorch keeps the graph of a failed backward call alive until the next call on the thread starts: the pass may still be around here)
    m.log.clear()
    m.zero_grad(set_to_none=True)
    m(x).sum().backward()                                                             # a clean pass: the old one is CLOSED, not run to its end
    assert m.log.count("kernels a") == 1 and m.log.count("kernels c") == 1 and m.log.count("closed") == 2, m.log
    assert torch.equal(m.a.grad, torch.full((3,), 18.0)) and m.active is None
    # a call that never reaches the pass's last nodes (autograd.grad over the last segment's parameters only): closed when its graph goes
    m3 = _ChainStub()
    o = m3(x)
    (g,) = torch.autograd.grad(o.sum(), [m3.c])
    del o
    gc.collect()
    assert m3.log == ["kernels c", "closed"] and m3.active is None, m3.log


def test_grad_segments_cover_the_completion_order():
    """the cut points of the chain are the engine's progress points; segments partition completion_order() in order"""
    import v_diffusion
    from v_diffusion.engine import UNetEngine
    for cfg in (dict(in_channels=3, hid_channels=32, out_channels=3, ch_multipliers=[1, 2, 2], num_res_blocks=2, apply_attn=[False, True, True],
                     num_classes=10), dict(in_channels=3, hid_channels=32, out_channels=6, ch_multipliers=[1, 2], num_res_blocks=1,
                                           apply_attn=[False, False], num_classes=5, multitags=True)):
        model = v_diffusion.UNet(**cfg)
        eng = UNetEngine(model)
        segs = eng.grad_segments()
        assert [n for _, names in segs for n in names] == eng.completion_order()
        pts = eng.progress_points()
        assert [b for b, _ in segs] == [p for p in pts if p not in ("in_conv.bias", None)] + [None]
        assert len(segs) == 2 * len(cfg["ch_multipliers"]) + 3            # out_conv, up levels, middle, down levels, the rest
        assert all(names[-1] == b for b, names in segs[:-1])
        assert segs[-1][1][:2] == ["in_conv.weight", "in_conv.bias"]
        assert all(k in segs[-1][1] for k in dict(model.named_parameters()) if k.startswith(("time_embed", "class_embed")) or ".norm" in k)
