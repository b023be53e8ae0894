"""The global `P` parameter object the reference's entry points mutate
(train/*_p.py: `P = Params()` built at import).  Only the fields of the retrieval path exist
(test mains set: test_pre_proc, cuda_device, image_input_size, test_batch_size, preload_net,
cnn_model, feature_size2d, embeddings_classify, feature_dim, num_classes, classif_model,
bn_model, regions_k -- test/*_test.py:44-56) and of siamese training (train/siamese_descriptor_p.py:62-101).
Unlike the reference nothing is read from disk at import."""


# Blocks of the base network that are NOT trained, lowest first (reference train/*_p.py:14-17,48: every training `Params` takes
# `untrained_blocks[cnn_model.lower()]`).  A block is a module with parameters of `features`; for a ResNet the stem has two (conv1,
# bn1) and every residual block is one, so the table freezes the stem and layers 1-3 and TRAINS layer4 (and conv5 of AlexNet).  The
# reference lists alexnet and resnet152; its own comment gives ResNet-50's layout (3, 4, 6, 3), the BASELINE's backbone.
UNTRAINED_BLOCKS = {
    'alexnet': 4,
    'resnet152': 2 + 3 + 8 + 36,
    'resnet50': 2 + 3 + 4 + 6,
    'resnet18': 2 + 2 + 2 + 2,
}


class Params(object):
    @property
    def untrained_blocks(self):
        """The table's entry for the CURRENT cnn_model unless a value was assigned (P.untrained_blocks = -1 freezes the whole
        trunk: descriptor-head-only training, this repo's round-2/3 benchmark configuration)."""
        v = self.__dict__.get('_untrained_blocks')
        if v is not None:
            return v
        try:
            return UNTRAINED_BLOCKS[str(self.cnn_model).lower()]       # an unknown backbone is an error, as in the reference (train/*_p.py:48)
        except KeyError:
            raise KeyError("no untrained_blocks entry for cnn_model %r (known: %s): assign P.untrained_blocks explicitly"
                           % (self.cnn_model, ", ".join(sorted(UNTRAINED_BLOCKS))))

    @untrained_blocks.setter
    def untrained_blocks(self, v):
        self.__dict__['_untrained_blocks'] = v

    def __init__(self, **overrides):
        self.cnn_model = 'AlexNet'
        self.cuda_device = 0
        self.image_input_size = (3, 224, 224)
        self.num_classes = 464
        self.feature_size2d = (6, 6)
        self.preload_net = ''
        self.bn_model = ''
        self.classif_model = ''
        self.test_batch_size = 64
        self.test_pre_proc = True
        self.test_trans = None                 # identity when test_pre_proc (images arrive normalised)
        self.embeddings_classify = False
        # extension (BASELINE configs[0] "AlexNet fc7"): descriptor = classifier[:6], the 4096-d activation behind the second ReLU of the
        # AlexNet classifier (model/ModelDefinition.py:31-37).  The reference only offers pool5-flat (9216-d) and the class scores.
        self.embeddings_fc7 = False
        self.feature_dim = 2048
        self.regions_k = 6                     # train/siamese_regions_p.py:98
        # inference entry points fold BatchNorm into the convolutions and fuse the bias/residual/ReLU
        # epilogues (model/nn_utils.fold_batch_norm): same function, ~1.4x images/s on the fp32 trunk
        self.fold_bn = True
        self.train_bn = False
        # descriptor slabs / similarity matrices larger than this go to the CPU in the reference
        # (2**30 there, utils/train_siamese.py:30-43); sized here for 288 GB of HBM3E
        self.embeddings_cuda_size = 64 * 2 ** 30
        self.log_file = None
        # --- siamese training (train/siamese_descriptor_p.py:62-101 of the reference) ---
        self.dataset_full = ''                 # a dataset folder (+ its `test` sub-folder) or a `synthetic:` spec: what `python -m train.<approach>` reads
        self.test_upfront = True               # evaluate before training (reference train/siamese_descriptor_p.py:56)
        self.train = True
        self.train_epochs = 20
        self.train_batch_size = 64
        self.train_micro_batch = 8
        self.train_lr = 1e-3
        self.train_momentum = 0.9
        self.train_weight_decay = 5e-4
        self.train_annealing = {}
        self.train_loss_avg = False
        self.train_loss2_avg = False
        self.train_loss2_alpha = 1.0
        self.train_loss_int = 10
        self.train_test_int = 0
        self.train_pre_proc = True
        self.train_trans = None
        self.triplet_margin = 0.1
        self.train_epoch_switch = 2          # semi-hard negatives before this epoch, hard ones after
        self.save_dir = None
        self.train_seed = 0
        self.train_prefix_ahead = 8          # frozen trunk prefix of this many consecutive mini-batches in one launch (1: every step launches its own); same bits
        self.train_prefix_cache = False      # frozen trunk prefix: look the features of resident training images up in an HBM table instead of recomputing them every step (same bits)
        self.train_head_shard = True         # data parallel: the descriptor head's Linear sharded by output features across the ranks (isx/shard_head.py)
        self.train_fused_head_sgd = True     # descriptor head: weight gradient + SGD update as one kernel (isx_head_sgd_step); False: torch's optimizer on a dW tensor
        for k, v in overrides.items():
            setattr(self, k, v)
