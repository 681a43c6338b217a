"""``python train_mean_teacher_3D.py --model unet_3D ...`` on MI355X.

Command-line drop-in for the reference's code/train_mean_teacher_3D.py: same flag names and defaults
(:31-62); ``num_classes`` is fixed to 2 as in the reference (:84).  The hot loop (:134-166) runs as the
fused HIP step; the consistency term is live from iteration 0 (no ``iter_num < 1000`` gate in 3-D).
"""
import argparse

import torch

parser = argparse.ArgumentParser()
parser.add_argument('--root_path', type=str, default='../data/BraTS2019', help='Name of Experiment')
parser.add_argument('--exp', type=str, default='BraTs2019_Mean_Teacher', help='experiment_name')
parser.add_argument('--model', type=str, default='unet_3D', help='model_name')
parser.add_argument('--max_iterations', type=int, default=30000, help='maximum epoch number to train')
parser.add_argument('--batch_size', type=int, default=4, help='batch_size per gpu')
parser.add_argument('--deterministic', type=int, default=1, help='whether use deterministic training')
parser.add_argument('--base_lr', type=float, default=0.01, help='segmentation network learning rate')
parser.add_argument('--patch_size', type=int, nargs=3, default=[96, 96, 96], help='patch size of network input')
parser.add_argument('--seed', type=int, default=1337, help='random seed')
# label and unlabel
parser.add_argument('--labeled_bs', type=int, default=2, help='labeled_batch_size per gpu')
parser.add_argument('--labeled_num', type=int, default=25, help='labeled data')
# costs
parser.add_argument('--ema_decay', type=float, default=0.99, help='ema_decay')
parser.add_argument('--consistency_type', type=str, default="mse", help='consistency_type')
parser.add_argument('--consistency', type=float, default=0.1, help='consistency')
parser.add_argument('--consistency_rampup', type=float, default=200.0, help='consistency_rampup')
# additions of this implementation
parser.add_argument('--hip_graph', type=int, default=0, help='capture the step in a hipGraph and replay it')


def main(argv=None):
    args = parser.parse_args(argv)
    args.num_classes = 2
    from mis_hip.train_common import run_training
    from networks.net_factory_3d import net_factory_3d

    def make_model():
        net = net_factory_3d(net_type=args.model, in_chns=1, class_num=args.num_classes)
        if net is None:
            raise SystemExit(f"unknown --model {args.model}")
        return net

    return run_training(args, make_model, label_dtype=torch.int64, cons_start_iter=0, save_ema=False,
                        snapshot_fmt="../model/{}_{}/{}")     # train_mean_teacher_3D.py:252


if __name__ == "__main__":
    print(main())
