"""GAN surface (mirrors reference ``satflow/models/gan/__init__.py:1-2``)."""
from .discriminators import GANLoss, NLayerDiscriminator, define_discriminator
from .generators import define_generator

__all__ = ["GANLoss", "NLayerDiscriminator", "define_discriminator", "define_generator"]
