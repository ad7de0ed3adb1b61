from .rpn import RPNWithIgnore, StandardRPNHead, build_proposal_generator  # noqa: F401
