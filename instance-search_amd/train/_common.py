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
