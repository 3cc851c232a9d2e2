for sk in 0 139264 1050624 69888; do echo "skew=$sk"; SPIRAL_ARENA_SKEW=$sk python tools/batch_query.py 2 3 4; done
