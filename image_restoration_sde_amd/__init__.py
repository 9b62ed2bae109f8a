"""image-restoration-sde_amd: the IR-SDE reverse-diffusion sampler of Algolzw/image-restoration-sde,
rebuilt for MI355X (gfx950): hand-written HIP kernels behind a C ABI (include/irsde_hip.h,
csrc/ -> libirsde_hip.so) with the reference's own Python interface on top:

    IRSDE              codes/utils/sde_utils.py:80-361
    ConditionalUNet    codes/config/deraining/models/modules/DenoisingUNet_arch.py:18-134
    ConditionalNAFNet  codes/config/deraining/models/modules/DenoisingNAFNet_arch.py:85-187 (Refusion)
    DenoisingSDE, denoising_sde.ConditionalUNet
                       codes/utils/sde_utils.py:373-593, codes/config/denoising-sde/models/modules/DenoisingUNet_arch.py
    DenoisingModel     codes/config/deraining/models/denoising_model.py (inference surface)
    latent.UNet / latent.ConditionalNAFNet / LatentDenoisingModel
                       codes/config/latent-dehazing/models/{modules/UNet_arch.py, modules/DenoisingNAFNet_arch.py,
                       latent_denoising_model.py} (encode once, sample in the latent, decode once)
    metrics            codes/utils/img_utils.py:136-234 + codes/data/util.py:177-198 (tensor2img / PSNR / SSIM / Y channel)
"""
from ._lib import IrsdeError, IrsdeLibraryError, build_library  # noqa: F401
from .denoising_model import DenoisingModel, ReverseSDEDenoisingModel, create_model, define_G  # noqa: F401
from .dist import gather_batch, sample_shard, sample_sharded, shard_bounds  # noqa: F401
from .sde import IRSDE  # noqa: F401
from .unet import ConditionalUNet  # noqa: F401
from .nafnet import ConditionalNAFNet  # noqa: F401
from . import denoising_sde  # noqa: F401
from .denoising_sde import DenoisingSDE  # noqa: F401
from . import metrics  # noqa: F401
from . import latent  # noqa: F401
from . import latent_bokeh  # noqa: F401
from .latent import LatentDenoisingModel  # noqa: F401

__all__ = ["IRSDE", "DenoisingSDE", "denoising_sde", "metrics", "latent", "latent_bokeh", "LatentDenoisingModel", "ConditionalUNet", "ConditionalNAFNet", "DenoisingModel", "ReverseSDEDenoisingModel", "create_model", "define_G", "build_library",
           "IrsdeError", "IrsdeLibraryError", "shard_bounds", "gather_batch", "sample_shard", "sample_sharded"]
