from .params import Params

P = Params(cnn_model='ResNet152', feature_size2d=(7, 7), feature_dim=2048)
