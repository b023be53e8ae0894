"""Evaluation entry point of the global-descriptor approach (reference
test/classif_finetune_test.py): same options, same printed result lines.

  python -m test.classif_finetune_test --dataset=<folder | synthetic:CLICIDE_video_224sq:n=100>
         --model=alexnet|resnet152|resnet50 [--weights=<file>] --device=<int> --classify=<bool>
         --batch=<int> [--dba=<int>] [--fc7=<bool>]
--fc7 is an extension (BASELINE configs[0]): 4096-d AlexNet fc7 descriptors = classifier[:6]; the reference offers pool5 / class scores.
"""
from __future__ import print_function

import sys

from train._common import prepare_for_inference
from train.classif_finetune import P, fc7_size, get_class_net, get_embeddings, labels, test_classif_net
from train.global_p import feature_sizes, flat_feature_sizes, image_sizes
from . import _common as C


def usage():
    C.usage_text(sys.argv[0], [
        C.O_DATASET, C.O_MODEL,
        '--weights=\t<file>\tThe filename containing weights of a network trained for sub-region classification.\n',
        C.O_DEVICE,
        '--classify=\t<bool>\tTrue/yes/y/1 if the classification feature should be tested. Otherwise, convolutional '
        'features are tested.\n', C.O_BATCH, C.O_DBA, C.O_SLAB,
        '--fc7=\t<bool>\t(extension) AlexNet only: test the 4096-d fc7 activation (classifier[:6]) instead of pool5.\n'])


def main(dataset_full, model, weights, device, classify, batch_size, dba, fc7=False, save_slab=None, gallery_slab=None):
    dataset_id = C.dataset_id_of(dataset_full)
    del labels[:]
    print('Loading and transforming train/test sets.')
    test_set, test_train_set = C.load_sets(dataset_full, labels, raw=(device >= 0))   # GPU runs ingest uint8 pixels and normalise on the device
    # the globals the retrieval functions read (reference :44-56)
    P.test_pre_proc = True
    P.cuda_device = device
    P.image_input_size = image_sizes[dataset_id]
    P.test_batch_size = batch_size
    P.preload_net = weights
    P.cnn_model = model
    P.feature_size2d = feature_sizes[model, image_sizes[dataset_id]]
    P.embeddings_classify = classify
    P.embeddings_fc7 = bool(fc7) and not classify
    out_size = len(labels) if classify else flat_feature_sizes[model, P.image_input_size]

    print('Testing network on dataset with ID {0}'.format(dataset_id))
    class_net = C.dp_sync_net(get_class_net())
    if P.embeddings_fc7:
        out_size = fc7_size(class_net.classifier)
    P.feature_dim = out_size
    prepare_for_inference(class_net, P)
    c, t = C.dp_classify(test_classif_net, class_net, test_set)
    print('Classification (TEST): {0} / {1} - acc: {2:.4f}'.format(c, t, float(c) / t))
    test_embeddings = C.dp_embeddings(get_embeddings, class_net, test_set, device, out_size)
    ref_embeddings, test_train_set = C.gallery_embeddings(get_embeddings, class_net, test_train_set, device, out_size, labels, save_slab, gallery_slab)
    return C.evaluate_retrieval(test_embeddings, ref_embeddings, test_set, test_train_set, device, labels, dba)


if __name__ == '__main__':
    C.run_cli(sys.argv[1:], usage,
              {'dataset': ('dataset', 'dataset'), 'model': ('model', 'model'), 'weights': ('file', 'initialization weights'),
               'device': ('int', 'device'), 'classify': ('bool', 'classify'), 'batch': ('int', 'batch'), 'dba': ('int', 'dba'), 'save-slab': ('path', 'slab file to write'), 'gallery-slab': ('file', 'slab file to read'), 'fc7': ('bool', 'fc7')},
              ('dataset', 'model', 'device', 'classify', 'batch'),
              lambda dataset, model, weights, device, classify, batch, dba, fc7=None, save_slab=None, gallery_slab=None:
              main(dataset, model, weights, device, classify, batch, dba, bool(fc7), save_slab, gallery_slab), P)
