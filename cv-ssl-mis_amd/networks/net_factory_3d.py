"""``net_factory_3d(net_type, in_chns, class_num)`` -- the reference's 3-D model factory surface.

Mirrors code/networks/net_factory_3d.py:10-41 (same signature/keys, module returned on the device,
unknown key -> ``None``).  ``unet_3D`` and ``vnet`` (SURVEY.md s.8a rows a3 / a3') and ``unetr`` (row n4; MONAI-based
in the reference: parity unpinned, see networks/unetr.py) and ``swinunetr`` (monai.networks.nets.SwinUNETR in the reference:
parity unpinned, see networks/swinunetr.py) are on the hand-written HIP hot path; attention_unet / voxresnet / nnUNet are out
of scope.
"""
from networks.swinunetr import SwinUNETR
from networks.unet_3D import unet_3D
from networks.unetr import UNETR
from networks.vnet import VNet

_OUT_OF_SCOPE = ("attention_unet", "voxresnet", "nnUNet")


def net_factory_3d(net_type="unet_3D", in_chns=1, class_num=2):
    if net_type == "unet_3D":
        net = unet_3D(n_classes=class_num, in_channels=in_chns).cuda()
    elif net_type == "vnet":
        net = VNet(n_channels=in_chns, n_classes=class_num, normalization='batchnorm', has_dropout=True).cuda()
    elif net_type == "unetr":                      # net_factory_3d.py:23-36 (in_channels fixed to 1 there too)
        net = UNETR(in_channels=1, out_channels=class_num, img_size=(96, 96, 96), feature_size=16, hidden_size=768,
                    mlp_dim=3072, num_heads=12, pos_embed='perceptron', norm_name='instance', conv_block=True,
                    res_block=True, dropout_rate=0.0).cuda()
    elif net_type == "swinunetr":                  # net_factory_3d.py:37-38 (no .cuda() there; the HIP net lives on the device)
        net = SwinUNETR(img_size=(64, 64, 64), in_channels=in_chns, out_channels=class_num, feature_size=48)
    elif net_type in _OUT_OF_SCOPE:
        raise NotImplementedError(
            f"net_type '{net_type}' is a valid reference key but not built on the HIP hot path yet "
            "(SURVEY.md s.8)")
    else:
        net = None
    return net
