"""Kaldi-style neural network with the reference's interface (neuralNetworks/nnet.py): `[nnet]` configuration
-> DNN; `train` = the optimisation schedule (held-out validation, learning-rate halving with rollback, layer-wise
growth, periodic checkpoints, state prior); `decode` = pseudo-log-likelihoods for Kaldi.

The reference keeps the whole schedule in one method around a tf.Session (nnet.py:80-244); here the engine
lifetime belongs to the Trainer / Decoder objects and the schedule is a small state object (`_Schedule`), one method
per decision the reference takes.  Printed lines and file names are the reference's.
"""
import itertools
import os
import shutil

import numpy as np

from ..dataparallel import init_from_env
from .classifiers import activation as act
from .classifiers.dnn import DNN
from .decoder import Decoder
from .trainer import CrossEnthropyTrainer, MicrobatchSelector

_NONLINEARITIES = ('relu', 'sigmoid', 'tanh', 'linear')  # nnet.py:48-62


def _activation_chain(conf):
    """Batchnorm -> nonlinearity -> L2Norm -> Dropout, each present only if configured; the flags are compared as
    the strings the config file holds (reference nnet.py:42-72)"""
    chain = act.Batchnorm(None) if conf['batch_norm'] == 'True' else None
    if conf['nonlin'] not in _NONLINEARITIES:
        raise Exception('unkown nonlinearity')
    chain = act.TfActivation(chain, conf['nonlin'])
    if conf['l2_norm'] == 'True':
        chain = act.L2Norm(chain)
    keep = float(conf['dropout'])
    return act.Dropout(chain, keep) if keep < 1 else chain


class Nnet(object):
    """a class for a neural network that can be used together with Kaldi"""

    def __init__(self, conf, input_dim, num_labels):
        """conf: ConfigParser with the [nnet] and [directories] sections (config/config_AURORA4.cfg:102-153);
        input_dim: dimension of the UNSPLICED features; num_labels: number of pdf-ids"""
        self.conf = dict(conf.items('nnet'))
        self.conf['savedir'] = '/'.join((conf.get('directories', 'expdir'), self.conf['name']))
        self.rank, self.world, _ = init_from_env()
        os.makedirs(os.path.join(self.conf['savedir'], 'training'), exist_ok=True)
        self.input_dim = (1 + 2 * int(self.conf['context_width'])) * input_dim  # after the +-context splice
        grows = int(self.conf['add_layer_period']) > 0
        # optional key compute_dtype = float32 (default, the reference's arithmetic) | float32_mfma | bfloat16 (_lib.DTYPES)
        self.dnn = DNN(num_labels, int(self.conf['num_hidden_layers']), int(self.conf['num_hidden_units']),
                       _activation_chain(self.conf), grows, compute_dtype=self.conf.get('compute_dtype', 'float32'))

    # ---- helpers shared with the schedule ----
    def _say(self, text):
        if self.rank == 0:
            print(text)

    def _barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()

    def train(self, dispenser):
        """Train the neural network on the batches of `dispenser` (schedule of reference nnet.py:80-244)"""
        _Schedule(self, dispenser).run()
        # the state prior counts ALL alignment targets, validation utterances included (nnet.py:241-244)
        counts = dispenser.compute_target_count().astype(np.float32)
        if self.rank == 0:
            np.save(os.path.join(self.conf['savedir'], 'prior.npy'), counts / counts.sum())
        self._barrier()

    def decode(self, reader, writer):
        """log(posterior / prior) of every utterance `reader` yields, written through `writer`
        (reference nnet.py:246-289)"""
        decoder = Decoder(self.dnn, self.input_dim, reader.max_input_length)
        try:
            decoder.restore(self.conf['savedir'] + '/final')
            decoder.set_prior(np.load(self.conf['savedir'] + '/prior.npy'))
            # The reference evaluates one utterance per session run (nnet.py:270-286).  Frames are independent in
            # this model, so utterances are grouped into forward passes of up to `decode_batch_frames` frames
            # (optional [nnet] key, default 8192; 0 = one utterance per pass) and written in their order.
            # The reference's flooring line discards its result (nnet.py:283): nothing is floored here either.
            budget = int(self.conf.get('decode_batch_frames', '8192'))
            group, frames = [], 0
            for utt_id, features in _utterances(reader):
                if group and frames + features.shape[0] > budget:
                    _write_group(decoder, writer, group)
                    group, frames = [], 0
                group.append((utt_id, features))
                frames += features.shape[0]
            if group:
                _write_group(decoder, writer, group)
        finally:
            decoder.close()
        writer.close()


def _utterances(reader):
    """one pass over the reader: it reports `looped` when it has wrapped around (nnet.py:272-277)"""
    while True:
        utt_id, features, looped = reader.get_utt()
        if looped:
            return
        yield utt_id, features


def _write_group(decoder, writer, group):
    likelihoods = decoder.decode_batch([features for _, features in group])
    for (utt_id, _), like in zip(group, likelihoods):
        writer.write_next_utt(utt_id, like)


class _Schedule(object):
    """One call of Nnet.train: owns the trainer and the position in the data."""

    def __init__(self, net, dispenser):
        self.net, self.conf, self.dispenser = net, net.conf, dispenser
        self.training_dir = net.conf['savedir'] + '/training/'
        n_valid = int(self.conf['valid_batches'])
        per_minibatch = self.conf['numutterances_per_minibatch']
        per_minibatch = dispenser.size if per_minibatch == '-1' else int(per_minibatch)
        # The packed feed (optional [nnet] key packed_feed, default True; needs this package's dispenser): every rank
        # reads only the utterances of ITS micro-batches, straight into one batch buffer, CMVN + splice happen in HBM,
        # and the next batch is produced while the GPU works on the current one.  Same batches, same order, same
        # arithmetic as trainer.update(*dispenser.get_batch()).
        self.packed = self.conf.get('packed_feed', 'True') == 'True' and getattr(dispenser, 'packed', False)
        self.select = MicrobatchSelector(per_minibatch, net.rank, net.world)
        # the first batches of the data are held out for validation, then cut off (nnet.py:88-96)
        if self.packed:
            self.valid = dispenser.next_packed(self.select, n_valid * dispenser.size) if n_valid else None
        else:
            held_out = [dispenser.get_batch() for _ in range(n_valid)]
            self.valid = (tuple(list(itertools.chain.from_iterable(part)) for part in zip(*held_out))
                          if held_out else None)
        dispenser.split()
        self.total_steps = int(dispenser.num_batches * int(self.conf['num_epochs']))
        # resume from the checkpoint at or below starting_step, at the matching position in the data (:101-108)
        start, every = int(self.conf['starting_step']), int(self.conf['check_freq'])
        self.step = start - start % every
        for _ in range(self.step):
            dispenser.skip_batch()
        # optional [nnet] key seed: weight initialisation and dropout masks reproducible from the configuration (the
        # reference seeds nothing; without the key the trainer draws a seed from the OS)
        seed = {'seed': int(self.conf['seed'])} if 'seed' in self.conf else {}
        self.trainer = CrossEnthropyTrainer(
            net.dnn, net.input_dim, dispenser.max_input_length, dispenser.max_target_length,
            float(self.conf['initial_learning_rate']), float(self.conf['learning_rate_decay']), self.total_steps,
            per_minibatch, **seed)
        if self.packed and not hasattr(self.trainer, 'update_packed'):
            raise TypeError("packed_feed needs a trainer with update_packed / evaluate_packed; set packed_feed = False "
                            "in [nnet] for %s" % type(self.trainer).__name__)
        self.best_loss = self.best_step = None
        self.retries = 0

    # ---- persistence: every rank holds the same state, rank 0 writes, all wait ----
    def _gather(self):
        """collective: the fp32 masters may be sharded over the ranks (mixed-precision sharded exchange); a trainer
        with the reference's plain interface has nothing to gather"""
        gather = getattr(self.trainer, "gather_parameters", None)
        if gather is not None:
            gather()

    def _on_rank0(self, write):
        """rank 0 writes, everybody learns whether it worked: a failure (disk full, parameters not gathered) raised on rank
        0 alone would leave the other ranks waiting in the barrier for ever"""
        error = None
        if self.net.rank == 0:
            try:
                write()
            except Exception as exc:  # noqa: BLE001  (re-raised below, on every rank)
                error = exc
        if self.net.world > 1:
            import torch
            import torch.distributed as dist
            flag = torch.tensor([0 if error is None else 1], dtype=torch.int32)
            if dist.get_backend() == "nccl":
                flag = flag.cuda()
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if error is None and int(flag.item()):
                raise RuntimeError("rank 0 failed to write the checkpoint (see its log)")
        if error is not None:
            raise error

    def save(self, name):
        self._gather()
        self._on_rank0(lambda: self.trainer.save_trainer(self.training_dir + name))

    def accept(self, loss):
        """the current model becomes the one to fall back to"""
        self.best_loss, self.best_step, self.retries = loss, self.step, 0
        self.save('validated')

    def _prefetch(self):
        self.dispenser.prefetch(self.select)

    def validate(self):
        loss = self.trainer.evaluate_packed(self.valid) if self.packed else self.trainer.evaluate(*self.valid)
        self.net._say('validation loss at step %d: %f' % (self.step, loss))
        return loss

    def fall_back(self):
        """worse than the validated model: rewind the data, reload it, halve the learning rate (nnet.py:180-199).
        Returns False when the retries are used up."""
        for _ in range(self.step - self.best_step):
            self.dispenser.return_batch()
        self.trainer.restore_trainer(self.training_dir + 'validated')
        self.trainer.halve_learning_rate()
        self.step = self.best_step
        if self.retries == int(self.conf['valid_retries']):
            self.net._say('the validation loss is worse, terminating training')
            return False
        self.net._say('the validation loss is worse, returning to the previously validated model with halved '
                      'learning rate')
        self.retries += 1
        return True

    def grow(self):
        """layer-wise growth (nnet.py:209-229): a new hidden layer every add_layer_period steps"""
        period, depth = int(self.conf['add_layer_period']), int(self.conf['num_hidden_layers'])
        if period <= 0 or self.step % period or self.step // period >= depth:
            return
        self.net._say('adding layer, the model now holds %d/%d layers' % (self.step // period + 1, depth))
        self._gather()  # (the control ops touch the fp32 masters: whole on every rank first)
        self.trainer.control_ops['add'].run()
        self.trainer.control_ops['init'].run()
        if self.valid is not None:
            self.accept(self.validate())

    def run(self):
        conf, trainer = self.conf, self.trainer
        if conf['visualise'] == 'True' and self.net.rank == 0:
            logdir = conf['savedir'] + '/logdir'
            if os.path.isdir(logdir):
                shutil.rmtree(logdir)
            trainer.start_visualization(logdir)
        try:
            trainer.initialize()
            if self.step > 0:
                trainer.restore_trainer(self.training_dir + 'step%d' % self.step)
            if self.valid is not None:
                self.accept(self.validate())
            adaptive = conf['valid_adapt'] == 'True'
            while self.step < self.total_steps:
                if self.packed:
                    # (the batch after this one is read while the GPU is busy with this one; nothing is read past the
                    # last step, and a rollback puts an unused prefetch back: BatchDispenser.return_batch)
                    ahead = self._prefetch if self.step + 1 < self.total_steps else None
                    loss = trainer.update_packed(self.dispenser.next_packed(self.select), overlap=ahead)
                else:
                    loss = trainer.update(*self.dispenser.get_batch())
                self.net._say('step %d/%d loss: %f' % (self.step, self.total_steps, loss))
                self.step += 1
                if self.valid is not None and self.step % int(conf['valid_frequency']) == 0:
                    current = self.validate()
                    if adaptive and current > self.best_loss:
                        if not self.fall_back():
                            break
                        continue
                    if adaptive:
                        self.accept(current)
                self.grow()
                if self.step % int(conf['check_freq']) == 0:
                    self.save('step%d' % self.step)
            self._gather()
            self._on_rank0(lambda: trainer.save_model(conf['savedir'] + '/final'))
        finally:
            trainer.close()
