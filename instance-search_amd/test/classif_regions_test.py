"""Evaluation entry point of the sub-region classifier approach (reference
test/classif_regions_test.py): descriptor = class scores at the best-classified location."""
from __future__ import print_function

import sys

from train._common import prepare_for_inference
from train.classif_regions import P, get_class_net, get_embeddings, labels, test_classif_net
from train.global_p import feature_sizes, image_sizes
from . import _common as C


def usage():
    C.usage_text(sys.argv[0], [
        C.O_DATASET, C.O_MODEL,
        '--weights=\t<file>\tThe filename containing weights of a network trained for sub-region classification.\n',
        C.O_DEVICE, C.O_DBA, C.O_SLAB])


def main(dataset_full, model, weights, device, dba, save_slab=None, gallery_slab=None):
    dataset_id = C.dataset_id_of(dataset_full)
    del labels[:]
    print('Loading and transforming train/test sets.')
    test_set, test_train_set = C.load_sets(dataset_full, labels, raw=(device >= 0))   # GPU runs ingest uint8 pixels and normalise on the device
    P.test_pre_proc = True
    P.cuda_device = device
    P.preload_net = weights
    P.cnn_model = model
    P.feature_size2d = feature_sizes[model, image_sizes[dataset_id]]
    P.bn_model = ''

    print('Testing network on dataset with ID {0}'.format(dataset_id))
    class_net = C.dp_sync_net(get_class_net())
    prepare_for_inference(class_net, P)
    c, t = C.dp_classify(test_classif_net, class_net, test_set)
    print('Classification (TEST): {0} / {1} - acc: {2:.4f}'.format(c, t, float(c) / t))
    test_embeddings = C.dp_embeddings(get_embeddings, class_net, test_set, device, len(labels))
    ref_embeddings, test_train_set = C.gallery_embeddings(get_embeddings, class_net, test_train_set, device, len(labels), labels, save_slab, gallery_slab)
    return C.evaluate_retrieval(test_embeddings, ref_embeddings, test_set, test_train_set, device, labels, dba)


if __name__ == '__main__':
    C.run_cli(sys.argv[1:], usage,
              {'dataset': ('dataset', 'dataset'), 'model': ('model', 'model'), 'weights': ('file', 'initialization weights'),
               'device': ('int', 'device'), 'dba': ('int', 'dba'), 'save-slab': ('path', 'slab file to write'), 'gallery-slab': ('file', 'slab file to read')},
              ('dataset', 'model', 'device'),
              lambda dataset, model, weights, device, dba, save_slab=None, gallery_slab=None: main(dataset, model, weights, device, dba, save_slab, gallery_slab), P)
