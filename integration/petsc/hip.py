# config/BuildSystem/config/packages/hip.py -- BuildSystem package behind --with-hip (PETSc 3.6.1).
#
# Modelled on config/BuildSystem/config/packages/cuda.py:1-166 (dependencies declared in setupDependencies, :30-35; the
# compiler-agreement checks :50-80, :123-152) and cusp.py:1-73 (getSearchDirectories, an extra check after the generic
# configureLibrary).  The generic machinery does the rest: config/package.py:571 configureLibrary finds include/petschip.h and
# lib/libpetschip.so under --with-hip-dir (or --with-hip-include / --with-hip-lib) and links `phip_init`;
# package.py:633-655 consistencyChecks turns self.double / self.complex / self.requires32bitint into configure errors using the
# precisionProvider / indexProvider that config/PETSc/Configure.py:90-101 hangs on every package object; and because this module
# is called `hip`, config/PETSc/Configure.py defines PETSC_HAVE_HIP for a package that was found.
#
# Written in the subset of Python shared by 2 and 3 (the BuildSystem itself needs Python 2, config/configure.py:13).
from __future__ import generators
import config.package
import os

class Configure(config.package.Package):
  def __init__(self, framework):
    config.package.Package.__init__(self, framework)
    self.functions        = ['phip_init', 'phip_mat_mult', 'phip_vec_mdot', 'phip_halo_begin']
    self.includes         = ['petschip.h']
    self.liblist          = [['libpetschip.so'], ['libpetschip.a']]
    self.double           = 1   # PetscScalar is double in the C ABI (include/petschip.h)
    self.complex          = 0   # real scalars only
    self.requires32bitint = 1   # PetscInt is int32_t in the C ABI
    self.cxx              = 0   # the boundary is plain C; the kernels are already compiled into libpetschip
    self.hipArch          = 'gfx950'
    self.validArchs       = ['gfx950']
    return

  def __str__(self):
    output  = config.package.Package.__str__(self)
    if self.found: output += '  Arch:     '+self.hipArch+'\n'
    return output

  def setupHelp(self, help):
    import nargs
    config.package.Package.setupHelp(self, help)
    help.addArgument('HIP', '-with-hip-arch=<arch>', nargs.Arg(None, 'gfx950', 'GPU architecture libpetschip was built for (gfx950 = MI355X)'))
    return

  def setupDependencies(self, framework):
    config.package.Package.setupDependencies(self, framework)
    self.setCompilers = framework.require('config.setCompilers',self)
    self.headers      = framework.require('config.headers',self)
    self.scalartypes  = framework.require('PETSc.options.scalarTypes', self)
    self.indexTypes   = framework.require('PETSc.options.indexTypes', self)
    self.languages    = framework.require('PETSc.options.languages',   self)
    return

  def getSearchDirectories(self):
    yield ''
    if 'PETSCHIP_DIR' in os.environ:
      yield os.environ['PETSCHIP_DIR']
    yield os.path.join('/opt','petschip')
    return

  def configureTypes(self):
    '''The C ABI fixes PetscScalar = double and PetscInt = int32_t; say so in this package's words before anything is compiled'''
    if not self.scalartypes.scalartype == 'real':
      raise RuntimeError('Must use real numbers with HIP (--with-scalar-type=real)')
    if not self.scalartypes.precision == 'double':
      raise RuntimeError('Must use double precision with HIP (--with-precision=double)')
    if not self.indexTypes.integerSize == 32:
      raise RuntimeError('Must use 32 bit indices with HIP (do not give --with-64-bit-indices)')
    if 'with-hip-arch' in self.argDB and self.argDB['with-hip-arch']:
      if not self.argDB['with-hip-arch'] in self.validArchs:
        raise RuntimeError('HIP Error: specified architecture invalid; libpetschip is built for '+' '.join(self.validArchs)+'.  Example: \'--with-hip-arch=gfx950\'')
      self.hipArch = self.argDB['with-hip-arch']
    return

  def checkSizeofTypes(self):
    '''The glue passes PetscInt* / PetscScalar* straight through as int32_t* / double*: check that the C compiler agrees (the role of
       cuda.py checkSizeofVoidP / checkNVCCDoubleAlign, where two compilers had to agree; here there is one compiler and a fixed ABI)'''
    self.log.write('Checking that int32_t and double have the sizes the libpetschip C ABI assumes\n')
    includes = '#include <stdint.h>\n#include <petschip.h>\n'
    body     = 'char a[(sizeof(int32_t) == 4 && sizeof(double) == 8 && sizeof(void*) == 8) ? 1 : -1]; a[0] = 0; (void)a;\n'
    oldFlags = self.compilers.CPPFLAGS
    self.compilers.CPPFLAGS += ' '+self.headers.toString(self.include)
    ok = self.checkCompile(includes, body)
    self.compilers.CPPFLAGS = oldFlags
    if not ok:
      raise RuntimeError('HIP Error: petschip.h does not compile with the C compiler, or int32_t / double / void* do not have the sizes its ABI assumes')
    return

  def configureLibrary(self):
    '''Calls the regular package configureLibrary and then the checks libpetschip needs'''
    self.configureTypes()
    config.package.Package.configureLibrary(self)
    self.checkSizeofTypes()
    self.addDefine('HIP_ARCH', '"'+self.hipArch+'"')
    return
