"""CVAE decoder parameterization, inference surface of
pyqg_generative/models/cvae_regression.py (:18-52 constructor, :114-118 generate,
:128-145 generate_latent_noise / predict_snapshot / predict_mean_snapshot)."""
from .cgan_regression import _LatentCNN


class CVAERegression(_LatentCNN):
    kind = 'vae'
    NET_NAMES = ('decoder',)

    def __init__(self, regression='None', folder='model', div=False, decoder_var='adaptive', device=0, **kw):
        if div:
            raise NotImplementedError('only div=False has a device path')
        self._set_regression(regression)
        self.div, self.decoder_var = div, decoder_var
        self._load(folder, device)          # needs decoder.pt (the encoder is training-only), net_mean.pt with regression != 'None'
