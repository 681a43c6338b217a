"""``python test_CNNVIT.py --root_path ../data/ACDC --exp ... --model unet`` on MI355X.

Command-line drop-in for the reference's code/test_CNNVIT.py (inference for the CNN + ViT methods: cross teaching, CNN-meets-ViT):
every case of ``test.list`` is predicted slice by slice at 224 x 224 (nearest resize in, arg-max, nearest resize back; :43-63) by
either the SwinUnet (``ViT_seg(config, img_size=[224, 224])`` + ``load_from(config)``, :97-99) or the ``net_factory`` CNN
(:101), and scored per foreground class with (dice, asd, hd) of the binarised masks (:33-39: medpy's ``dc`` / ``asd`` / ``hd``,
restated in utils/metrics.py; like medpy they raise on an empty mask).  The flags are the reference's (:22-31; the Swin flags
``--cfg`` / ``--opts`` reach the reference through the module-level parser of its net_factory, :13-74, on the same command line).

Differences, on purpose: the reference asks ``input("select model to test: enter 1 vit 2 unet")`` (:93) and loads a checkpoint
from its author's home directory (:107); here ``--choice`` answers the question without a terminal (absent: the same prompt) and
the checkpoint is ``<snapshot_path>/<Choice(choice)>`` -- the rule of the reference's commented line :106 -- unless
``--checkpoint`` names a file.  Prediction of a volume = val_2D.predict_slices (one upload, both resizes as device gathers,
batched forwards, one download).  SimpleITK is not in this image: predictions go to ``<case>_pred.npz``.
"""
import argparse
import os
import shutil

import numpy as np
import torch

parser = argparse.ArgumentParser()
parser.add_argument('--root_path', type=str, default='../data/ACDC', help='Name of Experiment')
parser.add_argument('--exp', type=str, default='ACDC/Cross_Pseudo_Mean_Teacher_4group', help='experiment_name')
parser.add_argument('--model', type=str, default='unet', help='model_name')
parser.add_argument('--num_classes', type=int, default=4, help='output channel of network')
parser.add_argument('--labeled_num', type=int, default=7, help='labeled data')
parser.add_argument('--cfg', type=str, default="../code/configs/swin_tiny_patch4_window7_224_lite.yaml")
parser.add_argument("--opts", default=None, nargs='+')
parser.add_argument('--choice', type=str, default=None, help="1 / ema: the SwinUnet, anything else: net_factory(--model)")
parser.add_argument('--checkpoint', type=str, default=None, help='state_dict file (default: the snapshot directory rule)')


def calculate_metric_percase(pred, gt):
    """(dice, asd, hd) of the binarised masks (:33-39; the reference calls the third ``hd95`` but asks medpy for ``hd``)."""
    from utils import metrics as metric
    p, g = pred > 0, gt > 0
    return metric.dc(p, g), metric.asd(p, g), metric.hd(p, g)


def test_single_volume(case, net, test_save_path, FLAGS):
    from dataloaders.dataset import read_case
    from val_2D import predict_slices
    image, label = read_case(os.path.join(FLAGS.root_path, "data", case))
    prediction = predict_slices(image, net, (224, 224)).astype(label.dtype)
    np.savez_compressed(os.path.join(test_save_path, case + "_pred.npz"), prediction=prediction)
    return tuple(calculate_metric_percase(prediction == c, label == c) for c in (1, 2, 3))


def Choice(choice):
    """checkpoint name of the model the user picked (:81-87)"""
    return {'1': '{}_best_model1.pth', '2': '{}_best_model2.pth'}.get(choice, '{}_best_ema_model.pth')


def build_net(FLAGS, choice):
    """'1' / 'ema': the SwinUnet at 224 x 224 (:96-99); anything else: the net_factory CNN (:100-101)."""
    if choice in ('ema', '1'):
        from config import get_config
        from networks.vision_transformer import SwinUnet as ViT_seg
        config = get_config(FLAGS)
        if config.MODEL.PRETRAIN_CKPT is not None and not os.path.exists(config.MODEL.PRETRAIN_CKPT):
            config.MODEL.PRETRAIN_CKPT = None
        net = ViT_seg(config, img_size=[224, 224], num_classes=FLAGS.num_classes).cuda()
        net.load_from(config)
        return net
    from networks.net_factory import net_factory
    return net_factory(net_type=FLAGS.model, in_chns=1, class_num=FLAGS.num_classes)


def Inference(FLAGS):
    with open(os.path.join(FLAGS.root_path, 'test.list')) as f:
        image_list = sorted(item.replace('\n', '').split(".")[0] for item in f.readlines())
    base = "../model/{}_{}/".format(FLAGS.exp, FLAGS.labeled_num)
    snapshot_path, test_save_path = base + FLAGS.model, base + FLAGS.model + "_predictions/"
    if os.path.exists(test_save_path):
        shutil.rmtree(test_save_path)
    os.makedirs(test_save_path)
    choice = FLAGS.choice if FLAGS.choice is not None else input("select model to test: enter 1 vit 2 unet")
    net = build_net(FLAGS, choice)
    save_mode_path = FLAGS.checkpoint or os.path.join(snapshot_path, Choice(choice).format(FLAGS.model))
    net.load_state_dict(torch.load(save_mode_path), False)       # strict=False, as the reference (:109)
    print("init weight from {}".format(save_mode_path))
    net.eval()
    totals = np.zeros((3, 3))                                    # [class][dice, asd, hd]
    for case in image_list:
        totals += np.asarray(test_single_volume(case, net, test_save_path, FLAGS))
    return [row for row in totals / len(image_list)]


if __name__ == '__main__':
    FLAGS = parser.parse_args()
    metric = Inference(FLAGS)
    print(metric)
    print((metric[0] + metric[1] + metric[2]) / 3)
