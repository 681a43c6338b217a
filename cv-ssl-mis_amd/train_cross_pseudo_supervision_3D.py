"""``python train_cross_pseudo_supervision_3D.py --model unet_3D|vnet ...`` on MI355X.

Command-line drop-in for the reference's code/train_cross_pseudo_supervision_3D.py (same flags and defaults,
:31-64): two students of the same architecture, each supervised on the labeled half and by the OTHER network's
arg-max pseudo labels (cross-entropy) on the unlabeled half (:149-185).  Runs as
mis_hip.step.CrossTeachingTrainer(pseudo_ce=True); model1 is re-initialised with
``kaiming_normal_init_weight`` and model2 with ``xavier_normal_init_weight`` as in the reference (:79-96, :108-109).
"""
import torch

from train_mean_teacher_3D import parser

parser.set_defaults(exp='BraTs2019_Cross_Pseudo_Supervision')


def main(argv=None):
    args = parser.parse_args(argv)
    args.num_classes = 2
    from mis_hip.train_common import run_cross_teaching
    from networks.net_factory_3d import net_factory_3d

    def make_model():
        net = net_factory_3d(net_type=args.model, in_chns=1, class_num=args.num_classes)
        if net is None:
            raise SystemExit(f"unknown --model {args.model}")
        return net

    from mis_hip.train_common import kaiming_normal_init_weight, xavier_normal_init_weight
    return run_cross_teaching(args, make_model, make_model, label_dtype=torch.int64, pseudo_ce=True,
                              init_fns=(kaiming_normal_init_weight, xavier_normal_init_weight))   # :108-109


if __name__ == "__main__":
    print(main())
