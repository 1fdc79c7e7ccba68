import sys, os, random
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import importlib.util, torch
spec = importlib.util.spec_from_file_location("fz", "tests/fuzz/fuzz_cpu_gpu.py"); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
torch.set_num_threads(1)
import qsparse_amd as qs
qs.set_qsparse_options(log_on_created=False, log_during_train=False)
rng = random.Random(4242)
for i in range(775):
    st = rng.getstate()
    if i < 774:
        fz.one_case(rng, i, dry=True)
rng.setstate(st)
desc, factory, shape, dtype = fz.build(rng)
print(desc)
for steps in (1, 2, 3):
    for nf in (None, (float("nan"), 1)):
        a = fz.run(factory, shape, dtype, "cpu", 4000 + 774, steps, steps, False, False, True, False, False, nf, False)
        b = fz.run(factory, shape, dtype, "cuda", 4000 + 774, steps, steps, False, False, True, False, False, nf, False)
        for (ka, va), (kb, vb) in zip(a, b):
            if ka.startswith("state") and va.is_floating_point():
                bad = ((va != vb) & ~(va.isnan() & vb.isnan())).sum().item()
                print(steps, nf, ka, "mismatch", bad, "nan count", int(va.isnan().sum()), int(vb.isnan().sum()))
