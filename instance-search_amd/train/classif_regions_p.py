from .params import Params

P = Params(cnn_model='AlexNet', feature_size2d=(6, 6), feature_dim=464)
