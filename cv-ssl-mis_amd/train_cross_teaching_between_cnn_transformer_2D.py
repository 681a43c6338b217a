"""``python train_cross_teaching_between_cnn_transformer_2D.py ...`` on MI355X (BASELINE config 5).

Command-line drop-in for the reference's code/train_cross_teaching_between_cnn_transformer_2D.py (same flags and
defaults, :46-105: ``--exp ACDC/Cross_Teaching_Between_CNN_Transformer --batch_size 16 --labeled_bs 8``):
model1 = ``net_factory(args.model)`` (UNet), model2 = ``ViT_seg(config, ...)`` + ``load_from`` (:169-172), both at
``--patch_size`` 224x224 -- or at 256x256 with ``--patch_size 256 256 --opts DATA.IMG_SIZE 256 MODEL.SWIN.WINDOW_SIZE 8``
(the reference's own override mechanism, config.py:194-195; BASELINE config 5's literal image size).  The loop body (:216-263) runs as the fused HIP cross-teaching step; under ``torchrun``
each rank owns its 16+16 shard and the only exchange is one RCCL all-reduce per model's flat gradient bucket.
"""
import os

from train_mean_teacher_2D import parser

parser.set_defaults(exp='ACDC/Cross_Teaching_Between_CNN_Transformer', batch_size=16, labeled_bs=8)


def main(argv=None):
    args = parser.parse_args(argv)
    from config import get_config
    from mis_hip.train_common import run_cross_teaching
    from networks.net_factory import net_factory
    from networks.vision_transformer import SwinUnet as ViT_seg
    config = get_config(args)
    if config.MODEL.PRETRAIN_CKPT is not None and not os.path.exists(config.MODEL.PRETRAIN_CKPT):
        config.MODEL.PRETRAIN_CKPT = None
    if list(args.patch_size) != [config.DATA.IMG_SIZE] * 2:
        raise SystemExit(f"--patch_size {args.patch_size} != DATA.IMG_SIZE {config.DATA.IMG_SIZE}: both networks run at "
                         "the SwinUnet's image size (224 with window 7; 256 needs --opts DATA.IMG_SIZE 256 "
                         "MODEL.SWIN.WINDOW_SIZE 8)")

    def make_model1():
        return net_factory(net_type=args.model, in_chns=1, class_num=args.num_classes)

    def make_model2():
        net = ViT_seg(config, img_size=args.patch_size, num_classes=args.num_classes).cuda()
        net.load_from(config)
        return net

    return run_cross_teaching(args, make_model1, make_model2)


if __name__ == "__main__":
    print(main())
