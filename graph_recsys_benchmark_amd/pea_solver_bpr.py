"""Plumbing CLI for BASELINE.json config 1 ("PEAGCN on MovieLens latest-small ... via peagcn_solver_bpr.py").

Same flag names and defaults as the reference's experiment scripts
    experiments/peagcn_solver_bpr.py:19-53  (peagat_solver_bpr.py / peasage_solver_bpr.py: identical flag sets,
                                             plus --num_heads for PEAGAT, experiments/peagat_solver_bpr.py:33)
the same three-dict assembly (:68-101) and the `update_graph_input` subclass idiom (:107-109); then, instead of the
reference's 30-epoch BaseSolver.run() (out of scope: training driver, checkpoints, dataset ETL), ONE pass of its step
sequence on the synthetic preset of the dataset the flags name:

    model.train(); loss(batch) [+ backward + optimizer step with --train_step true]   solvers.py:211-218
    model.eval()                                                                      solvers.py:227 / models/base.py:88-96
    metrics(model, dataset)                                                           solvers.py:33-104

and prints one JSON object.  The processed datasets are absent, so `--dataset/--dataset_name` select the
SyntheticHIN preset of the same shape.  Needs a GPU (no CPU fallback).

    python -m graph_recsys_benchmark_amd.pea_solver_bpr --model PEAGCN --dataset Movielens --dataset_name latest-small
"""
import argparse
import json
import random

import numpy as np
import torch

MODEL_TYPE = 'Graph'
LOSS_TYPE = 'BPR'
GRAPH_TYPE = 'hete'

PRESET_OF = {('Movielens', 'latest-small'): 'ml_small', ('Movielens', '25m'): 'ml25m_shaped', ('Yelp', ''): 'yelp_shaped',
             ('Yelp', 'yelp'): 'yelp_shaped'}


def build_parser():
    parser = argparse.ArgumentParser()
    # Dataset params
    parser.add_argument('--dataset', type=str, default='Movielens', help='')
    parser.add_argument('--dataset_name', type=str, default='latest-small', help='')
    parser.add_argument('--if_use_features', type=str, default='false', help='')
    parser.add_argument('--num_core', type=int, default=10, help='')
    parser.add_argument('--num_feat_core', type=int, default=10, help='')
    parser.add_argument('--sampling_strategy', type=str, default='random', help='')
    parser.add_argument('--entity_aware', type=str, default='false', help='')
    # Model params
    parser.add_argument('--dropout', type=float, default=0, help='')
    parser.add_argument('--emb_dim', type=int, default=64, help='')
    parser.add_argument('--num_heads', type=int, default=1, help='')
    parser.add_argument('--repr_dim', type=int, default=16, help='')
    parser.add_argument('--hidden_size', type=int, default=64, help='')
    parser.add_argument('--meta_path_steps', type=str, default='2,2,2,2,2,2,2,2,2', help='')
    parser.add_argument('--channel_aggr', type=str, default='att', help='')
    parser.add_argument('--entity_aware_coff', type=float, default=0.1, help='')
    # Train params
    parser.add_argument('--init_eval', type=str, default='true', help='')
    parser.add_argument('--num_negative_samples', type=int, default=4, help='')
    parser.add_argument('--num_neg_candidates', type=int, default=99, help='')
    parser.add_argument('--device', type=str, default='cuda', help='')
    parser.add_argument('--gpu_idx', type=str, default='0', help='')
    parser.add_argument('--runs', type=int, default=5, help='')
    parser.add_argument('--epochs', type=int, default=30, help='')
    parser.add_argument('--batch_size', type=int, default=1024, help='')
    parser.add_argument('--num_workers', type=int, default=12, help='')
    parser.add_argument('--opt', type=str, default='adam', help='')
    parser.add_argument('--lr', type=float, default=0.001, help='')
    parser.add_argument('--weight_decay', type=float, default=0.001, help='')
    parser.add_argument('--early_stopping', type=int, default=20, help='')
    parser.add_argument('--save_epochs', type=str, default='5,10,15,20,25', help='')
    parser.add_argument('--save_every_epoch', type=int, default=26, help='')
    parser.add_argument('--metapath_test', type=str, default='true', help='')
    # not in the reference: which of its three PEA scripts this invocation stands for, and how much to run
    parser.add_argument('--model', type=str, default='PEAGCN', choices=['PEAGCN', 'PEAGAT', 'PEASage'])
    parser.add_argument('--train_step', type=str, default='false', help="'true': loss.backward() + optimizer step too")
    parser.add_argument('--eval_users', type=int, default=0, help='evaluate only the first n test users (0 = all)')
    return parser


def assemble_args(args):
    """dataset_args / model_args / train_args exactly as experiments/peagcn_solver_bpr.py:68-101 builds them (minus the
    folder paths: nothing is written)."""
    device = 'cuda:{}'.format(args.gpu_idx)
    dataset_args = {
        'dataset': args.dataset, 'name': args.dataset_name,
        'if_use_features': args.if_use_features.lower() == 'true', 'num_negative_samples': args.num_negative_samples,
        'num_core': args.num_core, 'num_feat_core': args.num_feat_core,
        'cf_loss_type': LOSS_TYPE, 'type': GRAPH_TYPE,
        'sampling_strategy': args.sampling_strategy, 'entity_aware': args.entity_aware.lower() == 'true',
        'model': args.model
    }
    model_args = {
        'model_type': MODEL_TYPE,
        'if_use_features': args.if_use_features.lower() == 'true',
        'emb_dim': args.emb_dim, 'hidden_size': args.hidden_size,
        'repr_dim': args.repr_dim, 'dropout': args.dropout,
        'meta_path_steps': [int(i) for i in args.meta_path_steps.split(',')], 'channel_aggr': args.channel_aggr,
        'entity_aware': args.entity_aware.lower() == 'true',
        'entity_aware_coff': args.entity_aware_coff
    }
    if args.model == 'PEAGAT':
        model_args['num_heads'] = args.num_heads
    train_args = {
        'init_eval': args.init_eval.lower() == 'true',
        'num_negative_samples': args.num_negative_samples, 'num_neg_candidates': args.num_neg_candidates,
        'opt': args.opt, 'runs': args.runs, 'epochs': args.epochs, 'batch_size': args.batch_size,
        'weight_decay': args.weight_decay, 'device': device, 'lr': args.lr, 'num_workers': args.num_workers,
        'save_epochs': [int(i) for i in args.save_epochs.split(',')], 'save_every_epoch': args.save_every_epoch,
        'metapath_test': args.metapath_test.lower() == 'true'
    }
    return dataset_args, model_args, train_args


def main(argv=None, dataset=None):
    """Runs the sequence and returns the result dict (also printed as one JSON line when run as a script).
    `dataset`: an already built SyntheticHIN (tests share one); by default the preset the flags name."""
    from . import models, solvers
    from .utils import SyntheticHIN, update_pea_graph_input

    args = build_parser().parse_args(argv)
    if args.device == 'cpu' or not torch.cuda.is_available():
        raise RuntimeError('the HIP path needs a GPU (there is no CPU fallback)')
    dataset_args, model_args, train_args = assemble_args(args)
    key = (args.dataset, args.dataset_name if args.dataset == 'Movielens' else '')
    if key not in PRESET_OF:
        raise NotImplementedError('no synthetic preset for %s / %s' % (args.dataset, args.dataset_name))
    if dataset is None:
        dataset = SyntheticHIN(PRESET_OF[key], seed=2019)
    if dataset_args['entity_aware']:
        raise NotImplementedError('entity_aware batches need the entity lists of the processed dataset (absent)')
    steps = model_args['meta_path_steps']
    train_args['num_metapaths'] = len(steps)            # the first n metapaths of the dataset's table
    torch.cuda.set_device(int(args.gpu_idx))

    # seeds as solvers.py:123-127 sets them for run 1
    seed = 2019 + 1
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)

    base = {'PEAGCN': models.PEAGCNRecsysModel, 'PEAGAT': models.PEAGATRecsysModel,
            'PEASage': models.PEASageRecsysModel}[args.model]

    class PEARecsysModel(base):                          # experiments/peagcn_solver_bpr.py:107-109
        def update_graph_input(self, ds):
            return update_pea_graph_input(dataset_args, train_args, ds)

    model_args['num_nodes'] = dataset.num_nodes          # solvers.py:130-131
    model_args['dataset'] = dataset
    model = PEARecsysModel(**model_args).to(train_args['device'])
    opt = torch.optim.Adam(model.parameters(), lr=train_args['lr'], weight_decay=train_args['weight_decay'])

    batch = torch.from_numpy(dataset.bpr_batch(batch_size=train_args['batch_size'])).to(train_args['device'])
    model.train()                                        # solvers.py:211-218
    if args.train_step.lower() == 'true':
        opt.zero_grad()
        loss = model.loss(batch)
        loss.backward()
        opt.step()
    else:
        with torch.no_grad():
            loss = model.loss(batch)
    train_loss = float(loss.detach().cpu().item())

    model.eval()                                         # solvers.py:227
    if not hasattr(dataset, 'test_pos_unid_inid_map'):
        dataset.eval_split(num_users=args.eval_users or None)
    with torch.no_grad():
        hr, ndcg, auc, eval_loss = solvers.metrics(model, dataset, train_args['num_neg_candidates'])
    out = {'model': args.model, 'preset': dataset.preset, 'num_nodes': dataset.num_nodes, 'metapaths': len(steps),
           'parameters': sum(p.numel() for p in model.parameters()), 'train_loss': train_loss,
           'HR@5': float(hr[0]), 'HR@10': float(hr[5]), 'HR@20': float(hr[15]),
           'NDCG@5': float(ndcg[0]), 'NDCG@10': float(ndcg[5]), 'NDCG@20': float(ndcg[15]),
           'AUC': float(auc[0]), 'eval_loss': float(eval_loss[0]), 'eval_users': len(dataset.test_pos_unid_inid_map)}
    main.last_model = model                              # tests look at the model afterwards
    return out


if __name__ == '__main__':
    print(json.dumps(main()))
