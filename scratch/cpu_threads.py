import torch, time, sys, os
sys.path.insert(0,'.')
from oracle import cpu_ref as O
from differentiable_ransac_amd import synth
print('cores', os.cpu_count())
pair = synth.two_view_pair(0, 2000)
noise = synth.gumbel_noise((1024, 2000), seed=1)
for th in (1, 4, 8, 16, 32, 64):
    torch.set_num_threads(th)
    ts=[]
    for rep in range(2):
        t0=time.perf_counter()
        with torch.no_grad():
            idx, ret, _ = O.gumbel_topk(pair['logits'], noise, 1.0, 5)
            smp = O.gather_samples(pair['matches'], ret)
            t1=time.perf_counter()
            E, ok, _ = O.nister_5pt(smp)
            models = O.compact_models(E, ok)
            t2=time.perf_counter()
            scores, masks = O.msac_score(pair['matches'], models, 7.5e-4, chunk=2048)
            b = int(torch.argmax(torch.nan_to_num(scores, nan=-1.0)))
            t3=time.perf_counter()
        ts.append((t3-t0, t1-t0, t2-t1, t3-t2))
    print('threads', th, 'best total %.3f s (sample %.3f solve %.3f score %.3f) -> %.0f hyps/s' % (*min(ts), 1024/min(ts)[0]))
