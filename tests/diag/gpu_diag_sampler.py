"""GPU diagnostic: denoise + sampler parity against the golden fixtures (reference outputs)."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import audiodiffuser_amd as A
from audiodiffuser_amd.weights import generate_noise
from gpu_helpers import make_net, rel_err

G = np.load(os.path.join(ROOT, "tests/golden/hotpath_golden.npz"))
dev = torch.device("cuda")
for tag, cfg in (("tiny", A.config_tiny()), ("c1", A.config_c1())):
    B, L = (2, 256) if tag == "tiny" else (2, 2048)
    for dtype in ("fp32", "bf16"):
        net, w = make_net(cfg, dtype)
        diff = A.EluDiffusion(sigma_data=0.2)
        xn = generate_noise(7, B, L)
        for si, sg in enumerate((20.0, 1.5, 0.05)):
            y = diff.denoise_fn((xn * sg).cuda(), net=net, sigma=torch.tensor(sg), inference=True, cond_scale=1.0)
            print(f"{tag}/{dtype} denoise sigma={sg}: {rel_err(y.cpu(), torch.from_numpy(G[f'denoise_{tag}_{si}'])):.3e}", flush=True)
        sv = torch.tensor([3.0, 0.3])
        y = diff.denoise_fn((xn * sv[:, None, None]).cuda(), net=net, sigmas=sv.cuda(), inference=True, cond_scale=1.0)
        print(f"{tag}/{dtype} denoise vec: {rel_err(y.cpu(), torch.from_numpy(G[f'denoise_{tag}_vec'])):.3e}", flush=True)
        noise = generate_noise(40, B, L).cuda()
        sg18 = A.KarrasSchedule(0.002, 80.0, 7.0, 18)()
        sg50 = A.KarrasSchedule(0.002, 80.0, 7.0, 50)()
        sg12 = A.KarrasSchedule(0.002, 80.0, 7.0, 12)()
        for graph in (False, True):
            try:
                t0 = time.time()
                y = A.EDMSampler(s_churn=0.0, s_noise=1.0, num_steps=18, use_heun=True, use_graph=graph)(noise, fn=diff.denoise_fn, net=net, sigmas=sg18)
                torch.cuda.synchronize(); t1 = time.time()
                print(f"{tag}/{dtype} graph={graph} heun18: {rel_err(y.cpu(), torch.from_numpy(G[f'smp_heun18_{tag}_net_final'])):.3e} ({t1-t0:.2f}s)", flush=True)
                y = A.EDMAlphaSampler(alpha=1.0, num_steps=18, use_graph=graph)(noise, fn=diff.denoise_fn, net=net, sigmas=sg18)
                print(f"{tag}/{dtype} graph={graph} alpha18: {rel_err(y.cpu(), torch.from_numpy(G[f'smp_alpha18_{tag}_net_final'])):.3e}", flush=True)
                y = A.DPMSampler(cond_scale=1.0, order=3, num_steps=50, multisteps=True, x0_pred=True, log_time_spacing=False, use_graph=graph)(noise, fn=diff.denoise_fn, net=net, sigmas=sg50)
                print(f"{tag}/{dtype} graph={graph} dpm50: {rel_err(y.cpu(), torch.from_numpy(G[f'smp_dpm50_{tag}_net_final'])):.3e}", flush=True)
                inj = torch.stack([torch.randn((B, 1, L), generator=torch.Generator().manual_seed(9000 + i)) for i in range(12)]).cuda()
                y = A.EDMSampler(s_tmin=0.05, s_tmax=50.0, s_churn=40.0, s_noise=1.003, num_steps=12, use_graph=graph)(noise, fn=diff.denoise_fn, net=net, sigmas=sg12, injected_noise=inj)
                print(f"{tag}/{dtype} graph={graph} churn12: {rel_err(y.cpu(), torch.from_numpy(G[f'smp_churn12_{tag}_net_final'])):.3e}", flush=True)
                # timing of a second (graph replay / eager) run
                torch.cuda.synchronize(); t0 = time.time()
                y = A.EDMSampler(s_churn=0.0, s_noise=1.0, num_steps=18, use_heun=True, use_graph=graph)(noise, fn=diff.denoise_fn, net=net, sigmas=sg18)
                torch.cuda.synchronize(); print(f"    second heun18 run: {time.time()-t0:.3f}s", flush=True)
            except Exception:
                traceback.print_exc()
    if tag == "tiny":
        # mock-fn trajectories through the interface-compatibility branch (sampler arithmetic only)
        mock = lambda x, net=None, sigma=None, **kw: 0.5 * x
        noise = generate_noise(40, B, L).cuda()
        y = A.EDMSampler(s_churn=0.0, s_noise=1.0, num_steps=18)(noise, fn=mock, net=None, sigmas=sg18.cuda())
        print("mock heun18:", rel_err(y.cpu(), torch.from_numpy(G['smp_heun18_tiny_mock_final'])))
        y = A.DPMSampler(1.0, order=3, num_steps=50, multisteps=True, x0_pred=True, log_time_spacing=False)(noise, fn=mock, net=None, sigmas=sg50.cuda())
        print("mock dpm50:", rel_err(y.cpu(), torch.from_numpy(G['smp_dpm50_tiny_mock_final'])))
