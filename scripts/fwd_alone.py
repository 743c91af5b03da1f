"""The streaming decoder forward ALONE on the chip (one stream, nothing beside it): `python scripts/fwd_alone.py <items> <reps> [variant]`
under `rocprofv3 --kernel-trace --stats` gives the kernel's own duration, without the side stream's clock kernels it shares the CUs with
inside the one-call step.  variant = ltg_config.tuning (bit 26: the first form of the kernel)."""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")


def main():
    import helpers as Hh
    from ltgan.engine import CsrRows, Engine
    I, reps = int(sys.argv[1]), int(sys.argv[2])
    variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    B = 100
    eng = Engine(I, lr=1e-4, precision="bf16", seed=1)
    eng.cfg.tuning = variant
    rng = np.random.default_rng(0)
    X = Hh.random_history(rng, B, I, mean_nnz=18)
    dev = eng.device
    batch = CsrRows(torch.from_numpy(X.indptr.astype(np.int32)).to(dev), torch.from_numpy(X.indices.astype(np.int32)).to(dev), 0, B)
    acts = eng.new_acts(B)
    for _ in range(3):
        eng.forward(batch, acts, rng_step=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        eng.forward(batch, acts, rng_step=2 + r)
    e1.record()
    torch.cuda.synchronize()
    print("items %d variant %d: %.1f us per forward (enc-0 .. row statistics, one stream)" % (I, variant, e0.elapsed_time(e1) * 1e3 / reps))


if __name__ == "__main__":
    main()
