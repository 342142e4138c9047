from . import gate_points_message_passing  # noqa: F401
