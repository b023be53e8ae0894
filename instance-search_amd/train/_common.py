"""Shared pieces of the four per-approach modules: batch staging and net construction."""
import torch

from isx import backbones


# Raw ingest: a dataset may carry the DECODED images as (H,W,3) uint8 tensors instead of normalised
# fp32 tensors; stage_batch then ships the uint8 batch (a quarter of the bytes over PCIe) and applies
# ToTensor + Normalize on the GPU (libisx isx_images_u8_to_f32).  test/_common.load_sets fills this in.
RAW_INGEST = {"mean": None, "std": None}


def normalise_u8_batch(x_u8, device):
    """(B,H,W,3) uint8 -> (B,3,H,W) fp32, (x/255 - mean) / std; HIP kernel on the GPU, same arithmetic in torch on the CPU."""
    mean, std = RAW_INGEST["mean"], RAW_INGEST["std"]
    if mean is None:
        raise RuntimeError("uint8 images in the dataset but no mean/std registered (train._common.RAW_INGEST)")
    if device >= 0:
        from isx import ops
        x_u8 = x_u8 if x_u8.is_cuda else x_u8.pin_memory().cuda(non_blocking=True)
        return ops.images_u8_to_f32(x_u8.contiguous(), mean, std, channels_last=True)
    x = x_u8.permute(0, 3, 1, 2).float().div_(255.0)
    m = torch.tensor(list(mean)).view(1, -1, 1, 1)
    s_ = torch.tensor(list(std)).view(1, -1, 1, 1)
    return (x - m) / s_


def stage_batch(batch, trans, device):
    """Stack the (already normalised unless `trans` is given) images of a batch and move them."""
    if trans is None and batch and batch[0][0].dtype == torch.uint8:
        return normalise_u8_batch(torch.stack([im for im, _, _ in batch], 0), device)
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
