import sys, time, torch
sys.path.insert(0, "/root/repo")
from egohmr_amd import synthetic as syn
from egohmr_amd.factory import batch_to_device, build_synthetic_model
dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0)
b = batch_to_device(syn.make_batch(256, 4096, seed=100), dev)
fs = model.fused_sampler
outs = {}
for mode in (False, True, False, True):
    model.overlap_encoders = mode
    for _ in range(2):
        fs.invalidate(); st = fs.prepare(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fs.invalidate(); st = fs.prepare(b)
    torch.cuda.synchronize()
    print("overlap", mode, (time.perf_counter() - t0) / 5 * 1e3, "ms")
    outs[mode] = (st.h_img.clone(), st.h_oth.clone(), st.betas.clone())
print("equal:", all(torch.equal(a, b) for a, b in zip(outs[False], outs[True])))
