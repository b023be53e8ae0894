"""Per-dataset lookup tables and label parsers keyed by the dataset folder name -- the
configuration data of the reference's train/global_p.py:19-84, which its test entry points
index with `parse_dataset_id(--dataset)`.  `resnet50` rows are added (BASELINE's backbone)."""


def match_label_fou_clean2(x):
    """'.../<a>_<b>_rest.jpg' -> '<a><b>'"""
    parts = x.split('/')[-1].split('_')
    return parts[0] + parts[1]


def match_label_video(x):
    """'.../<label>-frame.jpg' -> '<label>'"""
    return x.split('/')[-1].split('-')[0]


def match_label_oxford(x):
    """'.../<label>_rest.jpg' -> '<label>'"""
    return x.split('/')[-1].split('_')[0]


# dataset id -> (n classes, label parser, mean/std file stem, input size)
_DATASETS = (
    ('CLICIDE',                 464, match_label_video,      'CLICIDE_224sq',   (3, 224, 224)),
    ('CLICIDE_max_224sq',       464, match_label_video,      'CLICIDE_224sq',   (3, 224, 224)),
    ('CLICIDE_video_227sq',     464, match_label_video,      None,              (3, 227, 227)),
    ('CLICIDE_video_224sq',     464, match_label_video,      'CLICIDE_224sq',   (3, 224, 224)),
    ('CLICIDE_video_384',       464, match_label_video,      'CLICIDE_384',     (3, 224, 224)),
    ('CLICIDE_video_448',       464, match_label_video,      'CLICIDE_448',     (3, 224, 224)),
    ('fourviere_clean2_224sq',  311, match_label_fou_clean2, 'fourviere_224sq', (3, 224, 224)),
    ('fourviere_clean2_384',    311, match_label_fou_clean2, 'fourviere_384',   (3, 224, 224)),
    ('fourviere_clean2_448',    311, match_label_fou_clean2, 'fourviere_448',   (3, 224, 224)),
    ('oxford5k_video_224sq',     17, match_label_oxford,     'oxford5k_224sq',  (3, 224, 224)),
    ('oxford5k_video_384',       17, match_label_oxford,     'oxford5k_384',    (3, 224, 224)),
)

image_sizes = dict((d[0], d[4]) for d in _DATASETS)
num_classes = dict((d[0], d[1]) for d in _DATASETS)
match_label_functions = dict((d[0], d[2]) for d in _DATASETS)
mean_std_files = dict((d[0], 'data/cli.txt' if d[3] is None else 'data/%s_train_ms.txt' % d[3]) for d in _DATASETS)

# (model, input size) -> spatial size of the last feature map / flattened descriptor width
feature_sizes = {('alexnet', (3, 224, 224)): (6, 6)}
flat_feature_sizes = {('alexnet', (3, 224, 224)): 256 * 6 * 6}
for _m in ('resnet152', 'resnet50'):
    feature_sizes[(_m, (3, 224, 224))] = (7, 7)
    feature_sizes[(_m, (3, 227, 227))] = (8, 8)
    flat_feature_sizes[(_m, (3, 224, 224))] = 2048
