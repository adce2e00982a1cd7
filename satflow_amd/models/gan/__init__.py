"""GAN surface (mirrors reference ``satflow/models/gan/__init__.py:1-2``)."""
from .discriminators import CloudGANDiscriminator, GANLoss, NLayerDiscriminator, PixelDiscriminator, define_discriminator
from .generators import define_generator

__all__ = ["GANLoss", "NLayerDiscriminator", "PixelDiscriminator", "CloudGANDiscriminator", "define_discriminator", "define_generator"]
