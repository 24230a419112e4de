"""Round 4: fp8 policies BETWEEN 'wide' (E2 + D4) and 'all' at the two trained operating points (32^3 / batch 256 and 64^3 / batch 64 shards,
256 samples each): policy 'all' with layers switched back to bf16 through VV_FP8_OFF.  IoU delta against the C oracle + ms per step.
python profiles/microbench/fp8_policy_mid.py [32|64]"""
import contextlib, json, os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
import numpy as np, torch
import voxvae
from voxvae import synthetic as syn, trained as tr
from oracle import c_oracle as co
DEV = 'cuda:0'
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = 256
if D == 64:
    cfg, ep, dp, info = tr.train_operating_point(voxel=64, latent=64, batch=32, pool=256, device=DEV, dtype='bf16', max_steps=3000)
    x = np.concatenate([syn.make_voxels(256, 64, seed=4321)[:192], syn.make_voxels(64, 64, seed=777)], axis=0); shard = 64
else:
    cfg, ep, dp, info = tr.train_operating_point(voxel=32, latent=64, device=DEV)
    x = np.concatenate([syn.make_voxels(256, 32, seed=4321)[:192], syn.make_voxels(64, 32, seed=777)], axis=0); shard = 256
eps = syn.make_eps(N, 64, seed=70)
ref = co.vae_eval_forward(cfg, ep, dp, x, x, eps)
iou_r = ref['tp'] / np.maximum(ref['tp'] + ref['fp'] + ref['fn'], 1)
print('oracle IoU %.4f' % iou_r.mean(), file=sys.stderr, flush=True)
xd, ed = torch.from_numpy(x).to(DEV), torch.from_numpy(eps).to(DEV)
import src.module.nolbo as nolbo
def run(label, dtype, policy, off):
    os.environ['VV_FP8_OFF'] = off
    voxvae.set_default_dtype(dtype); voxvae.set_default_device(DEV); voxvae.set_fp8_policy(policy)
    with contextlib.redirect_stdout(sys.stderr):
        m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
    m._encoder.set_weights_dict(ep); m._decoder.set_weights_dict(dp)
    ious = []
    for lo in range(0, N, shard):
        _, stats, _, _ = m.eval_forward_device(xd[lo:lo + shard].contiguous(), xd[lo:lo + shard].contiguous(), ed[lo:lo + shard].contiguous())
        s = stats.double().cpu().numpy(); ious.append(s[:, 1] / np.maximum(s[:, 1] + s[:, 2] + s[:, 3], 1))
    iou = np.concatenate(ious); diff = iou - iou_r
    xs, es = xd[:shard].contiguous(), ed[:shard].contiguous()
    for _ in range(10): m.eval_forward_device(xs, xs, es)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): m.eval_forward_device(xs, xs, es)
    torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / 50
    q = sorted(('E%d' % (int(k[1:]) + 1)) for k in m._enc_eng.packed if k.startswith('q') and k[1:].isdigit()) + sorted(('D%d' % (int(k[1:]) + 1)) for k in m._dec_eng.packed if k.startswith('q') and k[1:].isdigit())
    r = {'label': label, 'fp8_layers': q, 'iou_delta_mean': float(abs(diff.mean())), 'signed': float(diff.mean()), 'stderr': float(diff.std(ddof=1) / np.sqrt(N)),
         'max_per_sample': float(np.abs(diff).max()), 'ms_per_step_one_stream': round(ms, 4), 'batch': shard}
    print(json.dumps(r), flush=True)
run('bf16', 'bf16', 'wide', '')
run('wide', 'fp8', 'wide', '')
if D == 64:      # layer names at 64^3: E2 32->16 (direct fp8), E3 16->8, E4 8->4, E5 tail; D2 4->8, D3 8->16, D4 16->32 (direct fp8)
    run('wide + D3', 'fp8', 'all', 'E3,E4,E5,D2')
    run('wide + E3', 'fp8', 'all', 'E4,E5,D2,D3')
    run('wide + E3 + D3', 'fp8', 'all', 'E4,E5,D2')
    run('wide + E3 + D3 + D2', 'fp8', 'all', 'E4,E5')
    run('all but E5', 'fp8', 'all', 'E5')
else:
    run('wide + D3', 'fp8', 'all', 'E3,E4,E5,D2')
    run('wide + E3', 'fp8', 'all', 'E4,E5,D2,D3')
    run('wide + E3 + D3', 'fp8', 'all', 'E4,E5,D2')
    run('all but E5', 'fp8', 'all', 'E5')
run('all', 'fp8', 'all', '')
