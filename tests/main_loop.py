"""main.py:100-207 of the reference, transcribed statement by statement (the file itself cannot be imported: it needs
tensorboardX and time.clock), shared by tests/test_training_glue.py (environment side = the product kernels on the host
emulator) and tests/test_gpu_vs_live_reference.py (environment side = the HIP path, the reference loaded from
oracle/_ref/).  Everything that is not the environment is the reference's own code: Policy, RolloutStorage,
ACKTR.update, utils.get_vec_normalize; the masks the loop computes are checked row by row against the reference's own
acktr.utils helpers."""
import os
import types
from collections import deque

import numpy as np
import torch


def default_args(rot, num_processes=6, device="cpu"):
    size = (10, 10, 10)
    return types.SimpleNamespace(channel=4, container_size=size, pallet_size=10, enable_rotation=rot, num_processes=num_processes,
                                 num_steps=5, hidden_size=256, gamma=1.0, save_model=True, save_interval=1, save_dir="x",
                                 log_interval=1, algorithm="a2c", value_loss_coef=0.5, entropy_coef=0.01, invalid_coef=2.0,
                                 lr=7e-4, eps=1e-5, alpha=0.99, tensorboard=False, device=device)


def run(args, make_envs, get_possible_position, get_rotation_mask, data_path, updates=3):
    """The loop.  make_envs(args, device) -> the factory-shaped env object (main.py:63); the two mask helpers have the
    reference's signatures (acktr/utils.py:37,64).  The reference must already be importable (ref_shims.install)."""
    from acktr import algo, utils
    from acktr.envs import VecNormalize          # noqa: F401
    from acktr.model import Policy
    from acktr.storage import RolloutStorage
    from acktr.utils import get_possible_position as ref_gpp, get_rotation_mask as ref_grm
    env_name, custom, time_now = "Bpp-v0", "test", "now"
    KFAC = os.environ.get("BPP_TEST_KFAC", "0") == "1"
    torch.manual_seed(1)
    device = torch.device(args.device)
    envs = make_envs(args, device)                                                                            # main.py:63
    actor_critic = Policy(envs.observation_space.shape, envs.action_space,
                          base_kwargs={'recurrent': False, 'hidden_size': args.hidden_size, 'args': args})  # :80-82
    actor_critic.to(device)
    # main.py:105-110 (the 'acktr' branch; its 'a2c' branch omits args= and cannot run in the reference either).
    # acktr=KFAC: K-FAC itself where its eigendecomposition converges on this tiny batch, else the same update with RMSprop
    agent = algo.ACKTR(actor_critic, args.value_loss_coef, args.entropy_coef, args.invalid_coef, acktr=KFAC, args=args,
                       **({} if KFAC else dict(lr=args.lr, eps=args.eps, alpha=args.alpha, max_grad_norm=0.5)))
    rollouts = RolloutStorage(args.num_steps, args.num_processes, envs.observation_space.shape, envs.action_space,
                              actor_critic.recurrent_hidden_state_size, can_give_up=False,
                              enable_rotation=args.enable_rotation, pallet_size=args.container_size[0])    # :112-119
    obs = envs.reset()                                                                                        # :121
    location_masks = []
    for observation in obs:
        if not args.enable_rotation:
            box_mask = get_possible_position(observation, args.container_size)
        else:
            box_mask = get_rotation_mask(observation, args.container_size)
        location_masks.append(box_mask)
    location_masks = torch.FloatTensor(np.array(location_masks)).to(device)
    rollouts.obs[0].copy_(obs)
    rollouts.location_masks[0].copy_(location_masks)
    rollouts.to(device)
    episode_rewards = deque(maxlen=10)
    episode_ratio = deque(maxlen=10)
    import time
    start = time.time()
    j = 0
    while j < updates:                                                                                        # `while True`
        j += 1
        for step in range(args.num_steps):
            with torch.no_grad():
                value, action, action_log_prob, recurrent_hidden_states = actor_critic.act(
                    rollouts.obs[step], rollouts.recurrent_hidden_states[step], rollouts.masks[step], location_masks)
            location_masks = []
            obs, reward, done, infos = envs.step(action)
            for i in range(len(infos)):
                if 'episode' in infos[i].keys():
                    episode_rewards.append(infos[i]['episode']['r'])
                    episode_ratio.append(infos[i]['ratio'])
            for observation in obs:
                if not args.enable_rotation:
                    box_mask = get_possible_position(observation, args.container_size)
                    assert box_mask == ref_gpp(observation.cpu(), args.container_size)        # the reference's own helper
                else:
                    box_mask = get_rotation_mask(observation, args.container_size)
                    np.testing.assert_array_equal(box_mask, ref_grm(observation.cpu(), args.container_size))
                location_masks.append(box_mask)
            location_masks = torch.FloatTensor(np.array(location_masks)).to(device)
            masks = torch.FloatTensor([[0.0] if done_ else [1.0] for done_ in done])
            bad_masks = torch.FloatTensor([[0.0] if 'bad_transition' in info.keys() else [1.0] for info in infos])
            rollouts.insert(obs, recurrent_hidden_states, action, action_log_prob, value, reward, masks, bad_masks, location_masks)
        with torch.no_grad():
            next_value = actor_critic.get_value(rollouts.obs[-1], rollouts.recurrent_hidden_states[-1], rollouts.masks[-1]).detach()
        rollouts.compute_returns(next_value, False, args.gamma, 0.95, False)
        value_loss, action_loss, dist_entropy, prob_loss, graph_loss = agent.update(rollouts)
        rollouts.after_update()
        if args.save_model:
            if (j % args.save_interval == 0) and args.save_dir != "":
                torch.save([actor_critic.state_dict(), getattr(utils.get_vec_normalize(envs), 'ob_rms', None)],
                           os.path.join(data_path, env_name + time_now + ".pt"))
        if j % args.log_interval == 0 and len(episode_rewards) > 1:
            total_num_steps = (j + 1) * args.num_processes * args.num_steps
            end = time.time()
            print("Updates {}, num timesteps {}, FPS {} \\n"
                  "Last {} training episodes: mean/median reward {:.1f}/{:.1f}, min/max reward {:.1f}/{:.1f}\\n"
                  "The dist entropy {:.5f}, The value loss {:.5f}, the action loss {:.5f}\\n"
                  "The mean space ratio is {}\\n".format(j, total_num_steps, int(total_num_steps / (end - start)),
                                                        len(episode_rewards), np.mean(episode_rewards), np.median(episode_rewards),
                                                        np.min(episode_rewards), np.max(episode_rewards), dist_entropy, value_loss,
                                                        action_loss, np.mean(episode_ratio)))
    assert all(np.isfinite(float(v)) for v in (value_loss, action_loss, dist_entropy, prob_loss, graph_loss))
    path = os.path.join(data_path, env_name + time_now + ".pt")
    assert os.path.exists(path)
    saved = torch.load(path, weights_only=False)
    assert saved[1] is None and len(saved[0]) > 0
    return dict(episode_rewards=list(episode_rewards), episode_ratio=list(episode_ratio), envs=envs)
