"""TrackingTask under the reference's module path (envs/tasks/tracking_task.py); the class itself lives in task_base.py — target re-draw,
observation, reward and termination of this task are fused into the HIP step / reset kernels."""
from .task_base import TrackingTask  # noqa: F401
