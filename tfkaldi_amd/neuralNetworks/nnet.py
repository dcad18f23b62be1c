"""Kaldi-style neural network with the reference's interface (neuralNetworks/nnet.py): configuration ->
DNN, the training loop with validation / learning-rate halving / rollback / layer-wise growth /
checkpoints, and the decoding loop that writes pseudo-log-likelihoods for Kaldi."""
import itertools
import os
import shutil

import numpy as np

from ..dataparallel import init_from_env
from .classifiers import activation as act
from .classifiers.dnn import DNN
from .decoder import Decoder
from .trainer import CrossEnthropyTrainer


class Nnet(object):
    """a class for a neural network that can be used together with Kaldi"""

    def __init__(self, conf, input_dim, num_labels):
        """
        Args:
            conf: nnet configuration (a ConfigParser holding the [nnet] and [directories] sections,
                config/config_AURORA4.cfg:102-153)
            input_dim: network input dimension (unspliced features)
            num_labels: number of target labels
        """
        self.conf = dict(conf.items('nnet'))
        self.conf['savedir'] = conf.get('directories', 'expdir') + '/' + self.conf['name']
        self.rank, self.world, _ = init_from_env()
        for d in (self.conf['savedir'], self.conf['savedir'] + '/training'):
            os.makedirs(d, exist_ok=True)

        # the input dimension of the spliced features (reference nnet.py:39)
        self.input_dim = input_dim * (2 * int(self.conf['context_width']) + 1)

        # activation chain, built exactly as reference nnet.py:42-72 (string compares included)
        activation = act.Batchnorm(None) if self.conf['batch_norm'] == 'True' else None
        if self.conf['nonlin'] in ('relu', 'sigmoid', 'tanh', 'linear'):
            activation = act.TfActivation(activation, self.conf['nonlin'])
        else:
            raise Exception('unkown nonlinearity')
        if self.conf['l2_norm'] == 'True':
            activation = act.L2Norm(activation)
        if float(self.conf['dropout']) < 1:
            activation = act.Dropout(activation, float(self.conf['dropout']))

        # optional [nnet] key compute_dtype = float32 (default, the reference's arithmetic) | bfloat16
        self.dnn = DNN(num_labels, int(self.conf['num_hidden_layers']), int(self.conf['num_hidden_units']),
                       activation, int(self.conf['add_layer_period']) > 0,
                       compute_dtype=self.conf.get('compute_dtype', 'float32'))

    def _say(self, text):
        if self.rank == 0:
            print(text)

    def train(self, dispenser):
        """
        Train the neural network (control flow of reference nnet.py:80-244)

        Args:
            dispenser: a batchdispenser for training
        """
        conf = self.conf
        savedir = conf['savedir']
        # the validation set is read first and split off
        val_batches = [dispenser.get_batch() for _ in range(int(conf['valid_batches']))]
        if val_batches:
            val_data, val_labels = zip(*val_batches)
            val_data = list(itertools.chain.from_iterable(val_data))
            val_labels = list(itertools.chain.from_iterable(val_labels))
        else:
            val_data = val_labels = None
        dispenser.split()

        num_steps = int(dispenser.num_batches * int(conf['num_epochs']))

        # the saving point closest to the starting step, and the matching position in the data
        step = int(conf['starting_step']) - int(conf['starting_step']) % int(conf['check_freq'])
        for _ in range(step):
            dispenser.skip_batch()

        if conf['numutterances_per_minibatch'] == '-1':
            numutterances_per_minibatch = dispenser.size
        else:
            numutterances_per_minibatch = int(conf['numutterances_per_minibatch'])

        trainer = CrossEnthropyTrainer(
            self.dnn, self.input_dim, dispenser.max_input_length, dispenser.max_target_length,
            float(conf['initial_learning_rate']), float(conf['learning_rate_decay']), num_steps,
            numutterances_per_minibatch)

        if conf['visualise'] == 'True' and self.rank == 0:
            if os.path.isdir(savedir + '/logdir'):
                shutil.rmtree(savedir + '/logdir')
            trainer.start_visualization(savedir + '/logdir')

        def save_trainer(name):  # every rank holds the same state: rank 0 writes, all wait
            if self.rank == 0:
                trainer.save_trainer(savedir + '/training/' + name)
            self._barrier()

        try:
            trainer.initialize()
            if step > 0:
                trainer.restore_trainer(savedir + '/training/step' + str(step))

            if val_data is not None:
                validation_loss = trainer.evaluate(val_data, val_labels)
                self._say('validation loss at step %d: %f' % (step, validation_loss))
                validation_step = step
                save_trainer('validated')
                num_retries = 0

            while step < num_steps:
                batch_data, batch_labels = dispenser.get_batch()
                loss = trainer.update(batch_data, batch_labels)
                self._say('step %d/%d loss: %f' % (step, num_steps, loss))
                step += 1

                if step % int(conf['valid_frequency']) == 0 and val_data is not None:
                    current_loss = trainer.evaluate(val_data, val_labels)
                    self._say('validation loss at step %d: %f' % (step, current_loss))

                    if conf['valid_adapt'] == 'True':
                        if current_loss > validation_loss:
                            # worse: rewind the data, reload the validated model, halve the learning rate
                            for _ in range(step - validation_step):
                                dispenser.return_batch()
                            trainer.restore_trainer(savedir + '/training/validated')
                            trainer.halve_learning_rate()
                            step = validation_step
                            if num_retries == int(conf['valid_retries']):
                                self._say('the validation loss is worse, terminating training')
                                break
                            self._say('the validation loss is worse, returning to the previously validated '
                                      'model with halved learning rate')
                            num_retries += 1
                            continue
                        else:
                            validation_loss = current_loss
                            validation_step = step
                            num_retries = 0
                            save_trainer('validated')

                # layer-wise growth
                period = int(conf['add_layer_period'])
                if period > 0:
                    if step % period == 0 and step // period < int(conf['num_hidden_layers']):
                        self._say('adding layer, the model now holds %d/%d layers' % (
                            step // period + 1, int(conf['num_hidden_layers'])))
                        trainer.control_ops['add'].run()
                        trainer.control_ops['init'].run()
                        validation_loss = trainer.evaluate(val_data, val_labels)
                        self._say('validation loss at step %d: %f' % (step, validation_loss))
                        validation_step = step
                        save_trainer('validated')
                        num_retries = 0

                if step % int(conf['check_freq']) == 0:
                    save_trainer('step' + str(step))

            if self.rank == 0:
                trainer.save_model(savedir + '/final')
            self._barrier()
        finally:
            trainer.close()

        # the state prior (over ALL alignment targets, validation utterances included)
        prior = dispenser.compute_target_count().astype(np.float32)
        prior = prior / prior.sum()
        if self.rank == 0:
            np.save(savedir + '/prior.npy', prior)
        self._barrier()

    def decode(self, reader, writer):
        """
        compute pseudo likelihoods of the testing set (reference nnet.py:246-289)

        Args:
            reader: a feature reader object to read features to decode
            writer: a writer object to write likelihoods
        """
        decoder = Decoder(self.dnn, self.input_dim, reader.max_input_length)
        prior = np.load(self.conf['savedir'] + '/prior.npy')
        try:
            decoder.restore(self.conf['savedir'] + '/final')
            decoder.set_prior(prior)
            # The reference evaluates one utterance per session run (nnet.py:270-286).  Frames are independent
            # in this model, so utterances are grouped into forward passes of up to `decode_batch_frames`
            # frames (optional [nnet] key, default 8192; 0 = one utterance per pass) and written in order.
            budget = int(self.conf.get('decode_batch_frames', '8192'))
            pending, frames = [], 0

            def flush():
                # log(posterior / prior); the reference's flooring line discards its result (nnet.py:283),
                # so no flooring is applied there either
                for (uid, _), like in zip(pending, decoder.decode_batch([m for _, m in pending])):
                    writer.write_next_utt(uid, like)
                del pending[:]

            while True:
                utt_id, utt_mat, looped = reader.get_utt()
                if looped:
                    break
                if pending and frames + utt_mat.shape[0] > budget:
                    flush()
                    frames = 0
                pending.append((utt_id, utt_mat))
                frames += utt_mat.shape[0]
            if pending:
                flush()
        finally:
            decoder.close()
        writer.close()

    def _barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
