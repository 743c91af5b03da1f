"""Generator wrapper contract of the reference (Codes/generator.py:4-22, README.md:76-79):

    vae, item_probs, loss, params, p_dims, total_anneal_steps, anneal_cap = generator_VAECF(pro_dir)

`vae` exposes the four feed handles the training loop uses (input_ph, keep_prob_ph,
is_training_ph, anneal_ph -- MultiVAE.py:29-31,101-102); `item_probs` is the SOFTMAX output
(MultiVAE.py:143, although the reference calls it logits_var: Q1); `params` are the 8 tensors in
the order of MultiVAE.py:129-141 with TF shapes.  Instead of TF graph nodes the handles are small
symbolic objects that ltgan.session.Session.run understands; the arithmetic runs in the HIP
kernels behind include/ltg.h.
"""
from __future__ import annotations

from .dataset import count_items
from .engine import Engine


class Placeholder:
    """A feedable handle (tf.placeholder / tf.placeholder_with_default)."""

    def __init__(self, name, default=None):
        self.name, self.default = name, default

    def __repr__(self):
        return "<Placeholder %s>" % self.name


class Fetch:
    """A fetchable handle (a tensor or an op of the reference's graph)."""

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "<Fetch %s>" % self.name


class MultiVAE:
    """The recommender object of the wrapper contract (Base_Recommender/MultiVAE.py:95-230)."""

    def __init__(self, p_dims, lam=0.0, random_seed=98765, engine=None):
        self.p_dims = p_dims
        self.q_dims = p_dims[::-1]
        self.dims = self.q_dims + self.p_dims[1:]
        self.lam = lam                                   # L2 is disabled by the reference (lam=0.0, Q11)
        if lam != 0.0:
            raise NotImplementedError("lam != 0 is outside the reference's training path (generator.py:18)")
        self.random_seed = random_seed
        self.input_ph = Placeholder("input_ph")
        self.keep_prob_ph = Placeholder("keep_prob_ph", 0.75)    # MultiVAE.py:31 (Q3: on at inference)
        self.is_training_ph = Placeholder("is_training_ph", 0.0)  # MultiVAE.py:101
        self.anneal_ph = Placeholder("anneal_ph", 1.0)            # MultiVAE.py:102
        self.engine = engine

    def build_graph(self):
        """-> (softmax probabilities handle, neg_ELBO handle, params)  (MultiVAE.py:104-143)."""
        return Fetch("generator_out"), Fetch("g_vae_loss"), self.engine.generator_params_tf()


# The reference builds generator and discriminator into ONE TensorFlow default graph (train.py:127-136): the two factory calls
# share state without passing anything to each other.  The counterpart here is a process-wide "current engine".
_CURRENT = {"engine": None}


def current_engine():
    """The Engine of the last generator(...) call (None before the first / after reset_default_graph())."""
    return _CURRENT["engine"]


def reset_default_graph():
    """tf.reset_default_graph() (train.py:127): forget the current engine."""
    _CURRENT["engine"] = None


def _config_ini_defaults(path="config.ini"):
    """Discriminator sizes and learning rate of ./config.ini (train.py:359-377 reads it from the CWD) when the file exists;
    the reference's literal call `generator(pro_dir)` carries neither, they reach the graph through train_GAN's arguments."""
    import configparser
    import os
    cp = configparser.RawConfigParser()
    if not os.path.isfile(path) or not cp.read(path) or not cp.has_section("Long-Tail-GAN"):
        return {}
    out = {}
    try:
        out["h_sizes"] = tuple(int(cp.get("Long-Tail-GAN", k)) for k in ("h0_size", "h1_size", "h2_size", "h3_size"))
    except (configparser.NoOptionError, ValueError):
        pass
    try:
        out["lr"] = float(cp.get("Long-Tail-GAN", "LEARNING_RATE"))
    except (configparser.NoOptionError, ValueError):
        pass
    return out


def generator_VAECF(pro_dir, engine=None, **engine_kwargs):
    """Codes/generator.py:4-22 -- works with the reference's literal call `generator(pro_dir)` (train.py:130, test.py:79):
    discriminator sizes / learning rate default to ./config.ini when present (else config.ini's shipped defaults) and the
    engine becomes the process-wide current one that `discriminator(n_items, FEATURE_LEN, h0, h1, h2, h3)` finds.
    `engine` / keyword arguments override (item slab of a rank, precision, device ...)."""
    n_items = count_items(pro_dir)
    p_dims = [200, 600, n_items]          # VAECF recommended values (generator.py:13)
    total_anneal_steps = 20000            # generator.py:15
    anneal_cap = 0.2                      # generator.py:16
    if engine is None:
        import os
        kw = _config_ini_defaults()
        if os.environ.get("LTGAN_PRECISION"):          # decoder GEMM operands: "bf16" (default) or "fp32"
            kw["precision"] = os.environ["LTGAN_PRECISION"]
        kw.update(engine_kwargs)
        engine = Engine(n_items, p_dims=p_dims, seed=98765, **kw)
    _CURRENT["engine"] = engine
    vae = MultiVAE(p_dims, lam=0.0, random_seed=98765, engine=engine)
    probs_var, loss_var, params = vae.build_graph()
    return vae, probs_var, loss_var, params, p_dims, total_anneal_steps, anneal_cap


generator = generator_VAECF   # train.py:21 imports it under this name
