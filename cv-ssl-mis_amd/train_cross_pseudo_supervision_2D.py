"""``python train_cross_pseudo_supervision_2D.py --model unet ...`` on MI355X.

Command-line drop-in for the reference's code/train_cross_pseudo_supervision_2D.py (same flags and defaults,
:33-67; ``--patch_size`` 256 256): two UNet students, cross-entropy pseudo-supervision from the other network's
arg-max on the unlabeled half (:166-204).  Runs as mis_hip.step.CrossTeachingTrainer(pseudo_ce=True).
"""
import torch

from train_mean_teacher_2D import parser

parser.set_defaults(exp='ACDC/Cross_Pseudo_Supervision', patch_size=[256, 256], labeled_num=1)


def main(argv=None):
    args = parser.parse_args(argv)
    from mis_hip.train_common import run_cross_teaching
    from networks.net_factory import net_factory

    def make_model():
        net = net_factory(net_type=args.model, in_chns=1, class_num=args.num_classes)
        if net is None:
            raise SystemExit(f"unknown --model {args.model}")
        return net

    from mis_hip.train_common import kaiming_normal_init_weight, xavier_normal_init_weight
    return run_cross_teaching(args, make_model, make_model, label_dtype=torch.uint8, pseudo_ce=True,
                              init_fns=(kaiming_normal_init_weight, xavier_normal_init_weight))


if __name__ == "__main__":
    print(main())
