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


def generator_VAECF(pro_dir, engine=None, **engine_kwargs):
    """Codes/generator.py:4-22.  `engine` lets the caller share one Engine between the generator and
    the discriminator factories (the reference shares one TF default graph)."""
    n_items = count_items(pro_dir)
    p_dims = [200, 600, n_items]          # VAECF recommended values (generator.py:13)
    total_anneal_steps = 20000            # generator.py:15
    anneal_cap = 0.2                      # generator.py:16
    if engine is None:
        engine = Engine(n_items, p_dims=p_dims, seed=98765, **engine_kwargs)
    vae = MultiVAE(p_dims, lam=0.0, random_seed=98765, engine=engine)
    probs_var, loss_var, params = vae.build_graph()
    return vae, probs_var, loss_var, params, p_dims, total_anneal_steps, anneal_cap


generator = generator_VAECF   # train.py:21 imports it under this name
