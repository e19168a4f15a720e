import sys, warnings, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_gpu_trainer import make_opt
from test_gpu_fresh_orderings import _boosted_batch
from baseboostdepth_amd.trainer import Trainer
torch.backends.cudnn.deterministic = True
torch.backends.cudnn.benchmark = False
H, W, B = 96, 160, 4
A, Bm = [2, 1, 1, 0], [2, 2, 1, 1]
def run(graph, after, seq, fixed_noise):
    opt = make_opt(H, W, B, [0, 1, 2, 3], True)
    opt.step_graph, opt.graph_capture_after = graph, after
    torch.manual_seed(5)
    tr = Trainer(opt); tr.set_train()
    ls = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i, ms in enumerate(seq):
            b = _boosted_batch(ms, H, W, [0, 1, 2, 3], 60 + i, 0.3)
            if fixed_noise:
                g = torch.Generator(device="cuda").manual_seed(100 + i)
                b["noise"] = torch.randn(B, H, W, device="cuda", generator=g) * 1e-5
            _, losses = tr.train_step(b)
            ls.append(float(losses["loss"].detach()))
    torch.cuda.synchronize()
    return torch.cat([p.detach().flatten() for p in tr.parameters_to_train]), ls, tr
for fixed in (False, True):
    for seq in ([A, Bm, A, A, Bm, A], [A, A, A, A]):
        pe, le, _ = run(False, 0, seq, fixed)
        for after in (0, 1):
            pg, lg, trg = run(True, after, seq, fixed)
            print("fixed_noise", fixed, "seq", len(seq), "after", after, trg.graph_stats,
                  "loss rel", [abs(a - b) / abs(a) for a, b in zip(le, lg)], "param max rel", float((pe - pg).abs().max() / pe.abs().max()))
