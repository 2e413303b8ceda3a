"""Shared helpers for the parity tests (oracle = checker, HIP path = thing under test)."""
import numpy as np

from oracle import iaf_oracle as O

# fp32 parity bar (SURVEY.md section 8c / BASELINE.md section 4): max |y - y_fp64| on O(1) outputs
TOL_F32 = 2e-5


def set_hparams(cfg: O.ModelConfig, length=None, batch=None):
    """Point the global hparam singleton at ``cfg`` (what hp.set_hparam_yaml(case) would do)."""
    from pwv_amd.hparam import hparam as hp
    hp.set_hparam_yaml('default')
    m = hp.model
    m.dilations = [list(d) for d in cfg.dilations]
    m.filter_width = cfg.filter_width
    m.residual_channels = cfg.residual_channels
    m.dilation_channels = cfg.dilation_channels
    m.skip_channels = cfg.skip_channels
    m.condition_channels = cfg.condition_channels
    m.use_biases = cfg.use_biases
    m.use_skip_connection = cfg.use_skip_connection
    m.n_iaf = cfg.n_iaf
    m.normalize = cfg.normalize
    m.normalize_cond = cfg.normalize_cond
    m.normalize_wavenet = cfg.normalize_wavenet
    m.cond_upsample_method = cfg.cond_upsample_method
    m.shared_nets = cfg.shared_nets
    hp.signal.n_mels = cfg.n_mels
    hp.signal.hop_length = cfg.hop_length
    if length is not None:
        hp.generate.length = length
    if batch is not None:
        hp.generate.batch_size = batch
    return hp


def run_vocoder_hip(cfg, weights, mel, z, device, precision=None):
    """The HIP path through the reference-shaped host API: IAFVocoder(batch, length)(wav, mel, ...)."""
    import torch
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    set_hparams(cfg)
    store = VariableStore(device=device)
    store.load_dict(weights)
    n, length = z.shape[0], z.shape[1]
    model = IAFVocoder(batch_size=n, length=length, store=store, precision=precision)
    # enqueue-only + verify(): the parity tests must see the REQUESTED arithmetic or an exception -- not the call's own repair
    # (a rerun in exact fp32 would pass any parity bar); tests/test_safe_call.py covers the default, verified form
    out = model(None, torch.from_numpy(mel).to(device), is_training=False, z=torch.from_numpy(z).to(device), verify=False)
    model.verify()          # synchronises; raises PwvRangeError if the split-fp16 range guard fired
    return out.cpu().numpy()


def small_cfg(**kw):
    base = dict(dilations=[[1, 2, 4], [1, 2, 4, 8]], n_iaf=2)
    base.update(kw)
    return O.ModelConfig(**base)


def f16_storage_model(weights, cfg):
    """What the PWV_PREC_F16 build extension computes, restated for the fp64 oracle: the weights as its kernels
    hold them (fp16 after the exp2 scale folding of csrc/pwv_layer_common.h; dense / skip / postprocess1 plain
    fp16; everything at frame rate, biases and postprocess2 stay fp32) and the hook that rounds every activation
    the mode stores as fp16 (oracle.wavenet_forward(act_round=...)).  Returns (weights, act_round)."""
    k_f, k_g = np.float32(-2.8853900817779268), np.float32(-1.4426950408889634)

    def r16(a):
        return np.asarray(a).astype(np.float16).astype(np.float64)

    per_sample_cond = cfg.cond_upsample_method == 'transposed_conv'
    out = {}
    for name, v in weights.items():
        leaf = name.rsplit('/', 1)[1]
        in_stack = '/dilated_stack/' in name
        if in_stack and (leaf in ('filter', 'gate') or (per_sample_cond and leaf in ('gc_filter', 'gc_gate'))):
            k = k_f if leaf in ('filter', 'gc_filter') else k_g
            out[name] = r16(k * v.astype(np.float32)) / np.float64(k)
        elif (in_stack and leaf in ('dense', 'skip')) or leaf == 'postprocess1':
            out[name] = r16(v)
        else:
            out[name] = v
    return out, r16


def build_c_abi_smoke(out_path):
    """Compile examples/c_abi_smoke.c (a plain C99 client of the C ABI: raw hipMalloc pointers, no Python / torch)
    with gcc against the in-tree library.  Returns the command's CompletedProcess."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, 'parallel-wavenet-vocoder_amd')
    cmd = ['gcc', '-std=c99', '-Wall', '-D__HIP_PLATFORM_AMD__', os.path.join(root, 'examples', 'c_abi_smoke.c'),
           '-I' + os.path.join(root, 'include'), '-I/opt/rocm/include', '-L' + lib_dir, '-lpwv_hip', '-L/opt/rocm/lib',
           '-lamdhip64', '-lm', '-Wl,-rpath,' + lib_dir, '-Wl,-rpath,/opt/rocm/lib', '-o', out_path]
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
