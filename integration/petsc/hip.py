# config/BuildSystem/config/packages/hip.py -- BuildSystem package for --with-hip (pattern: cuda.py:1-166, cusp.py:1-73 of PETSc 3.6.1).
# Python 2, like the rest of that BuildSystem.  Defines PETSC_HAVE_HIP and links libpetschip; not runnable in this repository's image.
import config.package
import os

class Configure(config.package.Package):
  def __init__(self, framework):
    config.package.Package.__init__(self, framework)
    self.functions    = ['phip_init']
    self.includes     = ['petschip.h']
    self.liblist      = [['libpetschip.so']]
    self.double       = 1          # PetscScalar = double only
    self.cxx          = 0
    self.complex      = 0
    self.requires32bitint = 1      # PetscInt = int32 in the C ABI
    return

  def setupHelp(self, help):
    import nargs
    config.package.Package.setupHelp(self, help)
    help.addArgument('HIP', '-with-hipcc=<prog>', nargs.Arg(None, None, 'Specify the HIP compiler (default /opt/rocm/bin/hipcc)'))
    help.addArgument('HIP', '-with-hip-arch=<arch>', nargs.Arg(None, 'gfx950', 'GPU architecture to compile the kernels for'))
    return

  def getSearchDirectories(self):
    yield ''
    yield os.path.join('/opt', 'rocm')
    return

  def configureLibrary(self):
    config.package.Package.configureLibrary(self)
    if self.scalartypes.scalartype != 'real' or self.scalartypes.precision != 'double':
      raise RuntimeError('--with-hip needs --with-scalar-type=real --with-precision=double')
    if self.indexTypes.integerSize != 32:
      raise RuntimeError('--with-hip needs 32-bit PetscInt')
    self.addDefine('HAVE_HIP', 1)
    return
