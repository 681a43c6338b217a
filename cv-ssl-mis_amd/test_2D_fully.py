"""``python test_2D_fully.py --root_path ../data/ACDC --exp ... --model unet`` on MI355X.

Command-line drop-in for the reference's code/test_2D_fully.py: load
``../model/<exp>_<labeled_num>/<model>/<model>_best_model.pth``, predict every case of ``test.list`` slice by slice at
256 x 256 (nearest resize in, arg-max, nearest resize back; :41-63) and return the mean Dice of classes 1..3
(:33-39 keeps only the Dice).  Prediction of a volume = val_2D.predict_slices: one upload, both resizes as device
gathers, batched forwards, one download.  SimpleITK is not in this image: predictions go to ``<case>_pred.npz``.
"""
import argparse
import os
import shutil

import numpy as np
import torch

parser = argparse.ArgumentParser()
parser.add_argument('--root_path', type=str, default='../data/ACDC', help='Name of Experiment')
parser.add_argument('--exp', type=str, default='ACDC/Fully_Supervised', help='experiment_name')
parser.add_argument('--model', type=str, default='unet', help='model_name')
parser.add_argument('--num_classes', type=int, default=4, help='output channel of network')
parser.add_argument('--labeled_num', type=int, default=3, help='labeled data')


def calculate_metric_percase(pred, gt):
    from utils import metrics as metric
    return metric.dc(pred > 0, gt > 0)


def test_single_volume(case, net, test_save_path, FLAGS):
    from dataloaders.dataset import read_case
    from val_2D import predict_slices
    image, label = read_case(os.path.join(FLAGS.root_path, "data", case))
    prediction = predict_slices(image, net, (256, 256))
    np.savez_compressed(os.path.join(test_save_path, case + "_pred.npz"), prediction=prediction)
    return tuple(calculate_metric_percase(prediction == c, label == c) for c in (1, 2, 3))


def Inference(FLAGS):
    from networks.net_factory import net_factory
    with open(os.path.join(FLAGS.root_path, 'test.list')) as f:
        image_list = sorted(item.replace('\n', '').split(".")[0] for item in f.readlines())
    snapshot_path = "../model/{}_{}/{}".format(FLAGS.exp, FLAGS.labeled_num, FLAGS.model)
    if not os.path.isdir(snapshot_path):
        # the directory the Mean-Teacher / UA-MT 2-D training scripts write (train_mean_teacher_2D.py:328); the
        # reference keeps this spelling as the commented alternative at test_2D_fully.py:90
        labeled = "../model/{}_{}_labeled/{}".format(FLAGS.exp, FLAGS.labeled_num, FLAGS.model)
        if os.path.isdir(labeled):
            snapshot_path = labeled
    test_save_path = "../model/{}_{}/{}_predictions/".format(FLAGS.exp, FLAGS.labeled_num, FLAGS.model)
    if os.path.exists(test_save_path):
        shutil.rmtree(test_save_path)
    os.makedirs(test_save_path)
    net = net_factory(net_type=FLAGS.model, in_chns=1, class_num=FLAGS.num_classes)
    save_mode_path = os.path.join(snapshot_path, '{}_best_model.pth'.format(FLAGS.model))
    print(save_mode_path)
    net.load_state_dict(torch.load(save_mode_path))
    print("init weight from {}".format(save_mode_path))
    net.eval()
    totals = np.zeros(3)
    for case in image_list:
        totals += np.asarray(test_single_volume(case, net, test_save_path, FLAGS))
    return list(totals / len(image_list))


if __name__ == '__main__':
    print(Inference(parser.parse_args()))
