"""Host-side episode recorder for ONE sampled env of a batch (the observability role of the reference's
GIF recorder, ray.py:565-597,769-782 -- without its matplotlib figure layout): pulls that env's frame
to the host after each step and writes an animated GIF per finished episode with pillow.  Debug I/O,
off the hot path; nothing here runs unless the user calls it."""
import os

import numpy as np


class EpisodeRecorder:
    def __init__(self, venv, env_index=0, out_dir='renders', scale=4, every=1):
        if venv.obs_mode == 'state':
            raise ValueError('the recorder needs a pixel observation mode')
        self.venv, self.i, self.out_dir, self.scale, self.every = venv, int(env_index), out_dir, int(scale), int(every)
        self.frames, self.episode, self.saved = [], 0, []

    def _grab(self, tensor):
        f = tensor[self.i].cpu().numpy()
        return np.kron(f, np.ones((self.scale, self.scale, 1), dtype=np.uint8)) if self.scale > 1 else f

    def after_reset(self, obs):
        self.frames = [np.concatenate([self._grab(obs['observation']), self._grab(obs['desired_goal'])], axis=1)]

    def after_step(self, obs, done, info=None):
        """Call with step()'s results.  With auto-reset the row of a finished env already shows the next
        episode: its last frame comes from info['terminal_observation'] when the env keeps it."""
        fin = bool(done[self.i].item())
        goal = self.frames[0][:, self.frames[0].shape[1] // 2:] if self.frames else self._grab(obs['desired_goal'])
        if fin and info is not None and 'terminal_observation' in info:
            cur = self._grab(info['terminal_observation'])
        else:
            cur = self._grab(obs['observation'])
        if not fin or (info is not None and 'terminal_observation' in info):
            self.frames.append(np.concatenate([cur, goal], axis=1))
        if fin:
            path = self.save()
            self.episode += 1
            self.after_reset(obs)
            return path
        return None

    def save(self):
        if not self.frames or self.episode % self.every:
            return None
        from PIL import Image
        os.makedirs(self.out_dir, exist_ok=True)
        path = os.path.join(self.out_dir, 'env%d_E%d(%d).gif' % (self.i, self.episode, len(self.frames) - 1))
        imgs = [Image.fromarray(f) for f in self.frames]
        imgs[0].save(path, save_all=True, append_images=imgs[1:], duration=100, loop=0)
        self.saved.append(path)
        return path
