"""64^3 (BASELINE config 5's geometry): trained operating point, fp8 policies vs the C oracle.  One-off experiment."""
import os, sys, time
import numpy as np, torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
import voxvae
from voxvae import synthetic as syn, trained as tr
from oracle import c_oracle as co
DEV = 'cuda:0'
t0 = time.time()
cfg, ep, dp, info = tr.train_operating_point(voxel=64, latent=64, batch=32, pool=128, device=DEV, dtype='bf16', verbose=True, max_steps=1500)
print('fit', info['steps'], 'steps', round(time.time() - t0, 1), 's', info['history'][-1], flush=True)
n = 48
x = np.concatenate([syn.make_voxels(128, 64, seed=4321)[:32], syn.make_voxels(16, 64, seed=777)], axis=0)
eps = syn.make_eps(n, 64, seed=70)
t0 = time.time()
ref = co.vae_eval_forward(cfg, ep, dp, x, x, eps)
print('oracle', round(time.time() - t0, 1), 's', flush=True)
iou_r = ref['tp'] / np.maximum(ref['tp'] + ref['fp'] + ref['fn'], 1)
print('oracle IoU %.4f  logits [%.1f, %.1f]' % (iou_r.mean(), ref['logits'].min(), ref['logits'].max()))
xd, ed = torch.from_numpy(x).to(DEV), torch.from_numpy(eps).to(DEV)
import src.module.nolbo as nolbo
def run(dtype, policy):
    voxvae.set_default_dtype(dtype); voxvae.set_default_device(DEV); voxvae.set_fp8_policy(policy)
    m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
    m._encoder.set_weights_dict(ep); m._decoder.set_weights_dict(dp)
    _, z_act, _ = m._encode_latent(xd, ed)
    _, lg, st = m._dec_eng.forward(z_act, xd, want_logits=True)
    def step():
        return m.eval_forward_device(xd, xd, ed)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / 20
    s = st.double().cpu().numpy(); iou = s[:, 1] / np.maximum(s[:, 1] + s[:, 2] + s[:, 3], 1)
    lg = lg.cpu().numpy()
    print('%-5s %-5s IoU delta %+.2e max/sample %.2e rms dlogit %.4f  %.3f ms/step (B=%d)  q: %s' % (
        dtype, policy, iou.mean() - iou_r.mean(), np.abs(iou - iou_r).max(), np.sqrt(np.mean((lg - ref['logits']) ** 2)), ms, n,
        [k for k in list(m._enc_eng.packed) + list(m._dec_eng.packed) if k.startswith('q')]), flush=True)
run('bf16', 'wide'); run('fp8', 'wide'); run('fp8', 'all')
