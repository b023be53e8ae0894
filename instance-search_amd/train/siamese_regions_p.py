from .params import Params

P = Params(cnn_model='ResNet152', feature_size2d=(7, 7), feature_dim=2048, regions_k=6, train_micro_batch=1, train_lr=1e-4,
           train_weight_decay=0., train_loss2_avg=True, train_loss2_alpha=1.0)
