"""Evaluation entry point of the global siamese descriptor approach (reference
test/siamese_descriptor_test.py)."""
from __future__ import print_function

import sys

from train._common import prepare_for_inference
from train.global_p import feature_sizes, image_sizes
from train.siamese_descriptor import P, get_embeddings, get_siamese_net, labels
from . import _common as C


def usage():
    C.usage_text(sys.argv[0], [
        C.O_DATASET, C.O_MODEL,
        '--weights=\t<file>\tThe filename containing weights of a network trained as siamese descriptor.\n',
        C.O_DEVICE, '--feature-dim=\t<int>\tThe dimension of the descriptor.\n', C.O_BATCH, C.O_DBA, C.O_SLAB])


def main(dataset_full, model, weights, device, feature_dim, batch_size, dba, save_slab=None, gallery_slab=None):
    dataset_id = C.dataset_id_of(dataset_full)
    del labels[:]
    print('Loading and transforming train/test sets.')
    test_set, test_train_set = C.load_sets(dataset_full, labels, raw=(device >= 0))   # GPU runs ingest uint8 pixels and normalise on the device
    P.num_classes = len(labels)
    P.test_pre_proc = True
    P.cuda_device = device
    P.test_batch_size = batch_size
    P.preload_net = weights
    P.cnn_model = model
    P.feature_size2d = feature_sizes[model, image_sizes[dataset_id]]
    P.classif_model = ''
    P.feature_dim = feature_dim

    print('Testing network on dataset with ID {0}'.format(dataset_id))
    net = C.dp_sync_net(get_siamese_net())
    prepare_for_inference(net, P)
    test_embeddings = C.dp_embeddings(get_embeddings, net, test_set, device, net.feature_size)
    ref_embeddings, test_train_set = C.gallery_embeddings(get_embeddings, net, test_train_set, device, net.feature_size, labels, save_slab, gallery_slab)
    return C.evaluate_retrieval(test_embeddings, ref_embeddings, test_set, test_train_set, device, labels, dba)


if __name__ == '__main__':
    C.run_cli(sys.argv[1:], usage,
              {'dataset': ('dataset', 'dataset'), 'model': ('model', 'model'), 'weights': ('file', 'initialization weights'),
               'device': ('int', 'device'), 'feature-dim': ('int', 'feature-dim'), 'batch': ('int', 'batch'), 'dba': ('int', 'dba'), 'save-slab': ('path', 'slab file to write'), 'gallery-slab': ('file', 'slab file to read')},
              ('dataset', 'model', 'device', 'feature_dim', 'batch'),
              lambda dataset, model, weights, device, feature_dim, batch, dba, save_slab=None, gallery_slab=None:
              main(dataset, model, weights, device, feature_dim, batch, dba, save_slab, gallery_slab), P)
