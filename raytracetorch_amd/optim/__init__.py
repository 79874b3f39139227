from .goals import SpotSizeLoss
