"""Shared pieces of the four per-approach modules: batch staging and net construction."""
import torch

from isx import backbones


def stage_batch(batch, trans, device):
    """Stack the (already normalised unless `trans` is given) images of a batch and move them."""
    ims = [im if trans is None else trans(im) for im, _, _ in batch]
    x = torch.stack(ims, 0)
    if device >= 0:
        x = x.pin_memory().cuda(non_blocking=True) if not x.is_cuda else x
    return x


def test_transform(P):
    return None if P.test_pre_proc else P.test_trans


def base_model(P, pretrained=True):
    ctor = backbones.MODELS.get(P.cnn_model.lower())
    if ctor is None:
        raise ValueError('unknown cnn_model %r' % (P.cnn_model,))
    return ctor(pretrained=pretrained)


def load_weights(net, fname):
    if fname:
        net.load_state_dict(torch.load(fname, map_location='cpu'))
    return net


def prepare_for_inference(net, P):
    """eval mode (+ BatchNorm folding / fused epilogues when P.fold_bn): call once before get_embeddings."""
    from model.nn_utils import fold_batch_norm, set_net_train
    set_net_train(net, False)
    if getattr(P, 'fold_bn', False) and any(isinstance(m, torch.nn.BatchNorm2d) for m in net.features.modules()):
        dev = next(net.parameters()).device
        net.features = fold_batch_norm(net.features).to(dev)
    return net
