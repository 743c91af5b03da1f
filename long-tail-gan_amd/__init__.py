"""MI355X-native adversarial-NCF training path of Long-Tail-GAN (VAE-CF generator + MLP
discriminator + niche/popular sampling), a drop-in for the hot path of the reference's
Codes/train.py.  HIP kernels behind the C ABI of include/ltg.h; this package is the Python host.

The directory is named `long-tail-gan_amd` (not importable as written); import it through the
repo-root alias module:  `import ltgan`.
"""
__version__ = "0.1.0"
