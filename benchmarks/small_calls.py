import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import local_cases as cases
from fv3net_amd.local_mlp import LocalMlpModel, RnnModel
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
nz = 79
for ncol in (2304, 36864, 147456):
    st = cases.state(rng, nz, ncol, np.float64)
    d = {k: torch.from_numpy(v).to(dev) for k, v in st.items()}
    for label, model in (("dense-local", LocalMlpModel(cases.regressor(rng, st, nz, width=256, make=cases.product_makers()), device=dev)),
                         ("rnn", RnnModel(cases.precpd_rnn(rng, st, nz, channels=256, make=cases.product_makers()), device=dev))):
        for _ in range(3): model.predict(d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): model.predict(d)
        torch.cuda.synchronize()
        print(f"{label} ncol={ncol}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per call", flush=True)
